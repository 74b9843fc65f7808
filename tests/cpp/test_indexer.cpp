// C++ host test of include/gf_indexer.hpp: the planted-fusion known answer of
// SURVEY.md Appendix B through the C++ mirror of Indexer (needs a GPU).
#include <cstdio>
#include <random>

#include "gf_indexer.hpp"

using namespace genefuse;

static std::string rand_seq(std::mt19937& g, size_t n) {
  static const char b[] = "ACGT";
  std::string s(n, 'A');
  for (auto& c : s) c = b[g() & 3];
  return s;
}

static std::string rc(const std::string& s) {
  std::string r(s.rbegin(), s.rend());
  for (auto& c : r) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
  return r;
}

#define EXPECT(cond)                                             \
  do {                                                           \
    if (!(cond)) {                                               \
      printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);     \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main() {
  std::mt19937 g(3);
  Indexer::Contigs ref;
  ref["chr2"] = rand_seq(g, 3000);
  ref["7"] = rand_seq(g, 2500);
  std::vector<Fusion> fus(3);
  fus[0].m_gene = Gene{"A", "chr2", 100, 2100, false};
  fus[1].m_gene = Gene{"B", "chr7", 200, 2000, true};  // resolved by stripping "chr"
  fus[2].m_gene = Gene{"C", "chr9", 1, 100, false};    // missing chromosome
  Indexer ix = Indexer::with_loaded_ref(&ref, fus);
  ix.make_index();
  EXPECT(ix.m_fusion_seq.size() == 3 && ix.m_fusion_seq[2].empty());
  EXPECT(ix.m_fusion_seq[0] == ref["chr2"].substr(100, 2000));
  const std::string& g0 = ix.m_fusion_seq[0];
  const std::string& g1 = ix.m_fusion_seq[1];
  int p = 900, q = 600;
  while (g0[p + 1] == g1[q] || g1[q - 1] == g0[p]) { ++p; ++q; }
  std::string read = g0.substr(p - 74, 75) + g1.substr(q, 75);
  auto m = ix.map_read(read);
  EXPECT(m.size() == 2);
  EXPECT(m[0].seq_start == 0 && m[0].seq_end == 74 && m[0].start_gp.contig == 0 && m[0].start_gp.position == p - 74);
  EXPECT(m[1].seq_start == 75 && m[1].seq_end == 149 && m[1].start_gp.contig == 1 && m[1].start_gp.position == q - 75);
  EXPECT(ix.in_required_direction(m));  // left forward gene, right reversed gene (indexer.rs:578-590)
  auto mr = ix.map_read(rc(read));
  if (mr.size() != 2 || mr[0].seq_start != 75) {
    printf("rc read: %zu matches", mr.size());
    for (auto& x : mr) printf(" [%d-%d|%d:%d]", x.seq_start, x.seq_end, x.start_gp.contig, x.start_gp.position);
    printf(" expected [75-149|0:%d] [0-74|1:%d]\n", -(p + 75), -(q + 74));
  }
  EXPECT(mr.size() == 2 && mr[0].seq_start == 75 && mr[0].start_gp.position == -(p + 75));
  EXPECT(mr[1].start_gp.contig == 1 && mr[1].start_gp.position == -(q + 74));
  EXPECT(!ix.in_required_direction(mr));
  auto batch = ix.map_reads({read, std::string(150, 'N'), rc(read), g0.substr(300, 150)});
  EXPECT(batch.size() == 4 && batch[0].size() == 2 && batch[1].empty() && batch[2].size() == 2 && batch[3].empty());
  printf("OK\n");
  return 0;
}
