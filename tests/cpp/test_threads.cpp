// The boundary under threads: the reference calls Indexer::map_read from t-1 consumer threads on
// one `&self` (pescanner.rs:296-311, indexer.rs:252).  T = 8 std::threads call gf_map_reads /
// gf_map_reads_hits / gf_map_read on ONE index with different packs (batch route, small-call route
// and the streaming entry); every result must equal the serial one.  Prints the latency of an
// n = 1 gf_map_read and the aggregate rate of the threaded batch calls.  Needs a GPU.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gfmatch.h"

static std::string rand_seq(std::mt19937& g, size_t n) {
  static const char b[] = "ACGT";
  std::string s(n, 'A');
  for (auto& c : s) c = b[g() & 3];
  return s;
}

struct Pack {
  std::string bases;
  std::vector<int64_t> offsets;
  std::vector<int32_t> counts;        // serial answers
  std::vector<gf_seqmatch> matches;
  std::vector<gf_hit> hits;
  int64_t n() const { return (int64_t)offsets.size() - 1; }
};

#define EXPECT(cond)                                         \
  do {                                                       \
    if (!(cond)) {                                           \
      printf("FAIL %s:%d %s (%s)\n", __FILE__, __LINE__, #cond, gf_last_error()); \
      return 1;                                              \
    }                                                        \
  } while (0)

static bool same_match(const gf_seqmatch& a, const gf_seqmatch& b) {
  return a.seq_start == b.seq_start && a.seq_end == b.seq_end && a.position == b.position && a.contig == b.contig;
}

int main() {
  std::mt19937 g(11);
  const int G = 6, T = 8;
  std::vector<std::string> genes;
  for (int c = 0; c < G; ++c) genes.push_back(rand_seq(g, 20000));
  std::vector<const char*> gp;
  std::vector<int64_t> gl;
  for (auto& s : genes) { gp.push_back(s.data()); gl.push_back((int64_t)s.size()); }
  gf_index* ix = nullptr;
  EXPECT(gf_index_build(gp.data(), gl.data(), G, nullptr, &ix) == GF_OK);

  // packs: thread t gets its own reads — junction reads, single-gene reads, noise; ragged lengths
  std::vector<Pack> packs(T);
  for (int t = 0; t < T; ++t) {
    Pack& P = packs[t];
    const int n = 3000 + 137 * t;
    P.offsets.push_back(0);
    for (int r = 0; r < n; ++r) {
      std::string read;
      const int kind = (int)(g() % 4);
      const int L = 100 + (int)(g() % 60);
      if (kind == 0) {
        const int a = (int)(g() % G), b = (int)(g() % G);
        const int cut = 40 + (int)(g() % (L - 80));
        const size_t p = 200 + g() % 19000, q = 200 + g() % 19000;
        read = genes[a].substr(p - cut, cut) + genes[b].substr(q, L - cut);
      } else if (kind == 1) {
        read = genes[g() % G].substr(g() % 19800, L);
      } else {
        read = rand_seq(g, L);
      }
      P.bases += read;
      P.offsets.push_back((int64_t)P.bases.size());
    }
    P.counts.assign(n, 0);
    P.matches.assign(2 * (size_t)n, gf_seqmatch{});
    EXPECT(gf_map_reads(ix, P.bases.data(), P.offsets.data(), n, P.counts.data(), P.matches.data()) == GF_OK);
    P.hits.resize(n);
    int64_t nh = 0;
    EXPECT(gf_map_reads_hits(ix, P.bases.data(), P.offsets.data(), n, 1000 * t, P.hits.data(), n, &nh) == GF_OK);
    P.hits.resize((size_t)nh);
    int64_t two = 0;
    for (int r = 0; r < n; ++r) two += P.counts[r] == 2;
    EXPECT(two > 300 && nh >= two);
  }

  // T threads, each hammering its own pack through the three host entry points
  std::atomic<int> bad{0};
  std::atomic<long long> reads_done{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      const Pack& P = packs[t];
      const int64_t n = P.n();
      std::vector<int32_t> c(n);
      std::vector<gf_seqmatch> m(2 * (size_t)n);
      std::vector<gf_hit> h(n);
      for (int rep = 0; rep < 12; ++rep) {
        std::fill(c.begin(), c.end(), -1);
        if (gf_map_reads(ix, P.bases.data(), P.offsets.data(), n, c.data(), m.data()) != GF_OK) { bad++; return; }
        for (int64_t r = 0; r < n; ++r) {
          if (c[r] != P.counts[r]) { bad++; return; }
          for (int k = 0; k < c[r]; ++k)
            if (!same_match(m[2 * r + k], P.matches[2 * r + k])) { bad++; return; }
        }
        int64_t nh = 0;
        if (gf_map_reads_hits(ix, P.bases.data(), P.offsets.data(), n, 1000 * t, h.data(), n, &nh) != GF_OK) { bad++; return; }
        if (nh != (int64_t)P.hits.size() || memcmp(h.data(), P.hits.data(), (size_t)nh * sizeof(gf_hit)) != 0) { bad++; return; }
        reads_done += 2 * n;
        // one read at a time, like the reference's loop (the small-call route)
        for (int64_t r = rep; r < n; r += 97) {
          gf_seqmatch one[2];
          const int k = gf_map_read(ix, P.bases.data() + P.offsets[r], P.offsets[r + 1] - P.offsets[r], one);
          if (k != P.counts[r]) { bad++; return; }
          for (int j = 0; j < k; ++j)
            if (!same_match(one[j], P.matches[2 * r + j])) { bad++; return; }
        }
        // a sub-pack of 40 reads (small-call route with several reads)
        const int64_t s0 = (rep * 131) % (n - 40);
        if (gf_map_reads(ix, P.bases.data(), P.offsets.data() + s0, 40, c.data(), m.data()) != GF_OK) { bad++; return; }
        for (int64_t r = 0; r < 40; ++r) {
          if (c[r] != P.counts[s0 + r]) { bad++; return; }
          for (int k = 0; k < c[r]; ++k)
            if (!same_match(m[2 * r + k], P.matches[2 * (s0 + r) + k])) { bad++; return; }
        }
      }
    });
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  EXPECT(bad.load() == 0);
  printf("threads: %d threads x 12 rounds, %.2f M reads/s through host-buffer calls\n", T, reads_done.load() / dt / 1e6);

  // streams: every thread its own gf_stream, two packs in flight
  th.clear();
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      const Pack& P = packs[t];
      const int64_t n = P.n();
      gf_stream* s = nullptr;
      if (gf_stream_open(ix, n, (int64_t)P.bases.size(), 2, &s) != GF_OK) { bad++; return; }
      std::vector<gf_hit> h(n);
      int inflight = 0;
      for (int rep = 0; rep < 8; ++rep) {
        if (inflight == 2) {
          int64_t nh = 0;
          if (gf_stream_collect(s, h.data(), n, &nh) != GF_OK || nh != (int64_t)P.hits.size() ||
              memcmp(h.data(), P.hits.data(), (size_t)nh * sizeof(gf_hit)) != 0) { bad++; break; }
          inflight--;
        }
        if (gf_stream_submit(s, P.bases.data(), P.offsets.data(), n, 1000 * t) != GF_OK) { bad++; break; }
        inflight++;
      }
      while (inflight-- > 0) {
        int64_t nh = 0;
        if (gf_stream_collect(s, h.data(), n, &nh) != GF_OK || nh != (int64_t)P.hits.size() ||
            memcmp(h.data(), P.hits.data(), (size_t)nh * sizeof(gf_hit)) != 0) bad++;
      }
      // over capacity is an error, not a truncation
      if (gf_stream_submit(s, P.bases.data(), P.offsets.data(), n, 0) != GF_OK) bad++;
      if (gf_stream_submit(s, P.bases.data(), P.offsets.data(), n, 0) != GF_OK) bad++;
      if (gf_stream_submit(s, P.bases.data(), P.offsets.data(), n, 0) != GF_ERR_CAPACITY) bad++;
      gf_stream_close(s);
    });
  for (auto& x : th) x.join();
  EXPECT(bad.load() == 0);

  // latency of Indexer::map_read as the reference calls it: one read per call
  {
    const Pack& P = packs[0];
    gf_seqmatch one[2];
    for (int r = 0; r < 200; ++r) gf_map_read(ix, P.bases.data() + P.offsets[r], P.offsets[r + 1] - P.offsets[r], one);
    auto a = std::chrono::steady_clock::now();
    const int N = 3000;
    for (int r = 0; r < N; ++r) gf_map_read(ix, P.bases.data() + P.offsets[r], P.offsets[r + 1] - P.offsets[r], one);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / N;
    printf("latency: gf_map_read (n = 1) %.1f us per call\n", us);
  }
  EXPECT(gf_index_trim(ix) == GF_OK);
  {  // still usable after a trim
    const Pack& P = packs[1];
    std::vector<int32_t> c(P.n());
    std::vector<gf_seqmatch> m(2 * (size_t)P.n());
    EXPECT(gf_map_reads(ix, P.bases.data(), P.offsets.data(), P.n(), c.data(), m.data()) == GF_OK);
    EXPECT(c == P.counts);
  }
  gf_index_free(ix);
  {
    // multi-CSV mode runs its CSVs as concurrent jobs, each with an index of its own (fusion_scan.rs:103-110):
    // T threads build, map and free their own indexes side by side, three rounds each (the freed blocks of one
    // round are the next round's — and the other threads' — allocations); every thread's statistics and results
    // equal those of the same gene set built alone
    std::vector<std::vector<std::string>> sets(T);
    std::vector<gf_index_info> want(T);
    std::vector<std::vector<int32_t>> want_counts(T);
    const Pack& P = packs[0];
    auto build = [&](int t, gf_index** out) {
      std::vector<const char*> p2;
      std::vector<int64_t> l2;
      for (auto& s : sets[t]) { p2.push_back(s.data()); l2.push_back((int64_t)s.size()); }
      return gf_index_build(p2.data(), l2.data(), (int32_t)p2.size(), nullptr, out);
    };
    for (int t = 0; t < T; ++t) {
      for (int c = 0; c < 2 + t % 3; ++c) sets[t].push_back(c == 0 ? genes[t % G] : rand_seq(g, 5000 + 3000 * t));
      sets[t].push_back(sets[t][0].substr(100, 400));  // repeats: the one-pass build's side list is in use
      gf_index* one = nullptr;
      EXPECT(build(t, &one) == GF_OK);
      EXPECT(gf_index_info_get(one, &want[t]) == GF_OK);
      want_counts[t].assign(P.n(), 0);
      std::vector<gf_seqmatch> m(2 * (size_t)P.n());
      EXPECT(gf_map_reads(one, P.bases.data(), P.offsets.data(), P.n(), want_counts[t].data(), m.data()) == GF_OK);
      gf_index_free(one);
    }
    std::atomic<int> bad2{0};
    std::vector<std::thread> th2;
    for (int t = 0; t < T; ++t)
      th2.emplace_back([&, t] {
        for (int rep = 0; rep < 3; ++rep) {
          gf_index* mine = nullptr;
          if (build(t, &mine) != GF_OK) { bad2++; return; }
          gf_index_info got;
          std::vector<int32_t> c(P.n());
          std::vector<gf_seqmatch> m(2 * (size_t)P.n());
          if (gf_index_info_get(mine, &got) != GF_OK || got.n_keys != want[t].n_keys || got.n_sites != want[t].n_sites ||
              got.n_unique != want[t].n_unique || got.n_dupe_keys != want[t].n_dupe_keys ||
              got.n_high_keys != want[t].n_high_keys || got.n_dupe_sites != want[t].n_dupe_sites)
            bad2++;
          if (gf_map_reads(mine, P.bases.data(), P.offsets.data(), P.n(), c.data(), m.data()) != GF_OK || c != want_counts[t])
            bad2++;
          gf_index_free(mine);
        }
      });
    for (auto& x : th2) x.join();
    EXPECT(bad2.load() == 0);
    printf("concurrent index builds: %d threads x 3 rounds ok\n", T);
  }
  printf("OK\n");
  return 0;
}
