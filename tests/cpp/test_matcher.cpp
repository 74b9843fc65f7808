// include/gf_matcher.hpp against a case file written by tests/test_matcher.py:
//   line 1: n_contigs n_reads; then "name seq" per contig, then one read per line.
// Prints: "bloom B", "names a,b,c", per key "key K n c:p c:p ...", per read "none" | "panic", or
// "panic-build" when the constructor itself panics.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "gf_matcher.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1]);
  int nc = 0, nr = 0;
  f >> nc >> nr;
  std::map<std::string, std::string> contigs;
  for (int i = 0; i < nc; ++i) {
    std::string name, seq;
    f >> name >> seq;
    contigs[name] = seq;
  }
  std::vector<std::string> reads((size_t)nr);
  for (auto& r : reads) f >> r;
  try {
    genefuse::Matcher m(&contigs, reads);
    printf("bloom %u\n", m.bloom_bits);
    printf("names");
    for (auto& n : m.m_contig_names) printf(" %s", n.c_str());
    printf("\n");
    for (auto& kv : m.m_kmer_positions) {
      printf("key %d %zu", kv.first, kv.second.size());
      for (auto& s : kv.second) printf(" %d:%d", s.first, s.second);
      printf("\n");
    }
    for (auto& r : reads) {
      try {
        printf("%s\n", m.do_match(r) ? "match" : "none");
      } catch (const genefuse::MatcherPanic&) {
        printf("panic\n");
      }
    }
    try {
      auto kept = genefuse::remove_alignables(reads, &contigs);
      printf("kept %zu\n", kept.size());
    } catch (const genefuse::MatcherPanic&) {
      printf("kept panic\n");
    }
  } catch (const genefuse::MatcherPanic&) {
    printf("panic-build\n");
  }
  return 0;
}
