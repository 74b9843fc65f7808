// The boundary at the reference's own granularity, measured through the C ABI only.
//
// The reference maps one read per call inside packs of PACK_SIZE = 1000 pairs (common.rs:23) taken off a queue by
// t-1 consumer threads (pescanner.rs:296-311, 374-518).  A Rust host bound to include/gfmatch.h hands over one PACK per
// call instead; this program measures what a pack size buys: T std::threads on ONE index, each calling
// gf_map_reads_hits (one-shot host-buffer call) or driving its own gf_stream (submit / collect, depth 3) over packs of
// P pairs (2 P reads), from pageable (malloc) or pinned (gf_host_alloc) memory.  Prints one JSON object.
//
//   pack_sweep <genes.bin> <reads.bin> [seconds per cell] [--packs a,b,..] [--threads a,b,..] [--entries hits,stream]
//              [--mem pageable,pinned] [--check]
//   genes.bin: int32 n_genes | int64 len[n_genes] | the gene slices back to back
//   reads.bin: int64 n_reads | int64 offsets[n_reads + 1] | bases
// (both written by tools/bench_pack_sweep.py).  --check compares every cell's hit count per pack with a serial call.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gfmatch.h"

static std::vector<long long> parse_list(const char* s) {
  std::vector<long long> v;
  while (*s) {
    v.push_back(atoll(s));
    while (*s && *s != ',') ++s;
    if (*s == ',') ++s;
  }
  return v;
}

static std::vector<std::string> parse_names(const char* s) {
  std::vector<std::string> v;
  std::string cur;
  for (; *s; ++s) {
    if (*s == ',') { v.push_back(cur); cur.clear(); } else cur += *s;
  }
  if (!cur.empty()) v.push_back(cur);
  return v;
}

static bool read_all(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

struct Cell {
  std::string entry, mem;
  long long pack_pairs;
  int threads;
  double reads_per_s, us_per_call;
  long long calls, hits;
  bool ok;
};

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: pack_sweep genes.bin reads.bin [seconds] [--packs ..] [--threads ..] [--entries ..] [--mem ..] [--check]\n");
    return 2;
  }
  double secs = 0.4;
  std::vector<long long> packs = {1000, 4000, 16000, 64000, 256000, 1000000};
  std::vector<long long> threads = {1, 4, 8, 16};
  std::vector<std::string> entries = {"hits", "stream"}, mems = {"pageable", "pinned"};
  bool check = false;
  for (int a = 3; a < argc; ++a) {
    if (!strcmp(argv[a], "--packs") && a + 1 < argc) packs = parse_list(argv[++a]);
    else if (!strcmp(argv[a], "--threads") && a + 1 < argc) threads = parse_list(argv[++a]);
    else if (!strcmp(argv[a], "--entries") && a + 1 < argc) entries = parse_names(argv[++a]);
    else if (!strcmp(argv[a], "--mem") && a + 1 < argc) mems = parse_names(argv[++a]);
    else if (!strcmp(argv[a], "--check")) check = true;
    else secs = atof(argv[a]);
  }
  // ---- inputs ----
  FILE* fg = fopen(argv[1], "rb");
  FILE* fr = fopen(argv[2], "rb");
  if (!fg || !fr) { fprintf(stderr, "cannot open inputs\n"); return 2; }
  int32_t n_genes = 0;
  if (!read_all(fg, &n_genes, 4) || n_genes <= 0) return 2;
  std::vector<int64_t> gl((size_t)n_genes);
  if (!read_all(fg, gl.data(), 8 * (size_t)n_genes)) return 2;
  std::vector<std::string> genes((size_t)n_genes);
  std::vector<const char*> gp;
  for (int c = 0; c < n_genes; ++c) {
    genes[(size_t)c].resize((size_t)gl[(size_t)c]);
    if (!read_all(fg, &genes[(size_t)c][0], (size_t)gl[(size_t)c])) return 2;
    gp.push_back(genes[(size_t)c].data());
  }
  fclose(fg);
  int64_t n = 0;
  if (!read_all(fr, &n, 8) || n <= 0) return 2;
  std::vector<int64_t> offsets((size_t)n + 1);
  if (!read_all(fr, offsets.data(), 8 * ((size_t)n + 1))) return 2;
  const int64_t nb = offsets[(size_t)n];
  char* pageable = (char*)malloc((size_t)nb + 64);
  if (!pageable || !read_all(fr, pageable, (size_t)nb)) return 2;
  fclose(fr);
  gf_index* ix = nullptr;
  if (gf_index_build(gp.data(), gl.data(), n_genes, nullptr, &ix) != GF_OK) {
    fprintf(stderr, "gf_index_build: %s\n", gf_last_error());
    return 1;
  }
  char* pinned = (char*)gf_host_alloc(nb + 64);
  int64_t* pinned_off = (int64_t*)gf_host_alloc(8 * (n + 1));
  if (!pinned || !pinned_off) { fprintf(stderr, "gf_host_alloc failed\n"); return 1; }
  memcpy(pinned, pageable, (size_t)nb);
  memcpy(pinned_off, offsets.data(), 8 * ((size_t)n + 1));

  // serial answers: hits per pack of the smallest unit (for --check, the hit count of pack [p0, p0 + np) is a prefix difference)
  std::vector<int64_t> hit_prefix;
  if (check) {
    std::vector<gf_hit> h((size_t)n);
    int64_t nh = 0;
    if (gf_map_reads_hits(ix, pageable, offsets.data(), n, 0, h.data(), n, &nh) != GF_OK) { fprintf(stderr, "%s\n", gf_last_error()); return 1; }
    hit_prefix.assign((size_t)n + 1, 0);
    for (int64_t k = 0; k < nh; ++k) hit_prefix[(size_t)h[(size_t)k].read_id + 1] += 1;
    for (int64_t r = 0; r < n; ++r) hit_prefix[(size_t)r + 1] += hit_prefix[(size_t)r];
  }

  std::vector<Cell> cells;
  for (const std::string& entry : entries)
    for (const std::string& mem : mems)
      for (long long pp : packs) {
        const int64_t np = 2 * pp;  // reads per pack
        if (np > n) continue;
        const int64_t n_packs = n / np;
        for (long long T : threads) {
          const char* src = mem == "pinned" ? pinned : pageable;
          const int64_t* off = mem == "pinned" ? pinned_off : offsets.data();
          std::atomic<long long> reads_done{0}, calls_done{0}, hits_done{0};
          std::atomic<int> bad{0}, go{0};
          std::string err;
          std::mutex err_mu;
          auto fail_here = [&](const char* what) { std::lock_guard<std::mutex> lk(err_mu); if (err.empty()) err = std::string(what) + ": " + gf_last_error(); bad++; };
          std::atomic<bool> stop{false};
          std::vector<std::thread> th;
          for (int t = 0; t < (int)T; ++t)
            th.emplace_back([&, t] {
              std::vector<gf_hit> h((size_t)np);
              gf_stream* s = nullptr;
              const int depth = 3;
              if (entry == "stream") {
                int64_t max_bytes = 0;
                for (int64_t q = 0; q < n_packs; ++q) max_bytes = std::max<int64_t>(max_bytes, off[(q + 1) * np] - off[q * np]);
                if (gf_stream_open(ix, np, max_bytes, depth, &s) != GF_OK) { fail_here("gf_stream_open"); return; }
              }
              // warm: one call per thread before the clock starts (arenas, lanes, workspaces)
              int64_t nh = 0;
              int64_t pk = (t * 7) % n_packs;
              if (entry == "hits") {
                if (gf_map_reads_hits(ix, src, off + pk * np, np, pk * np, h.data(), np, &nh) != GF_OK) fail_here("warm gf_map_reads_hits");
              } else {
                if (gf_stream_submit(s, src, off + pk * np, np, pk * np) != GF_OK || gf_stream_collect(s, h.data(), np, &nh) != GF_OK) fail_here("warm stream");
              }
              go++;
              while (go.load() < (int)T && !bad.load()) std::this_thread::yield();
              long long my_reads = 0, my_calls = 0, my_hits = 0;
              std::vector<int64_t> inflight_pack;
              while (!stop.load(std::memory_order_relaxed) && !bad.load()) {
                pk = (pk + 1) % n_packs;
                if (entry == "hits") {
                  if (gf_map_reads_hits(ix, src, off + pk * np, np, pk * np, h.data(), np, &nh) != GF_OK) { fail_here("gf_map_reads_hits"); break; }
                  if (check && nh != hit_prefix[(size_t)((pk + 1) * np)] - hit_prefix[(size_t)(pk * np)]) { fail_here("hit count differs from the serial call"); break; }
                  my_hits += nh;
                  my_reads += np;
                  my_calls += 1;
                } else {
                  if ((int)inflight_pack.size() == depth) {
                    if (gf_stream_collect(s, h.data(), np, &nh) != GF_OK) { fail_here("gf_stream_collect"); break; }
                    const int64_t q = inflight_pack.front();
                    inflight_pack.erase(inflight_pack.begin());
                    if (check && nh != hit_prefix[(size_t)((q + 1) * np)] - hit_prefix[(size_t)(q * np)]) { fail_here("hit count differs from the serial call"); break; }
                    my_hits += nh;
                    my_reads += np;
                    my_calls += 1;
                  }
                  if (gf_stream_submit(s, src, off + pk * np, np, pk * np) != GF_OK) { fail_here("gf_stream_submit"); break; }
                  inflight_pack.push_back(pk);
                }
              }
              while (s && !inflight_pack.empty()) {  // (drained outside the clock: not counted)
                gf_stream_collect(s, h.data(), np, &nh);
                inflight_pack.erase(inflight_pack.begin());
              }
              if (s) gf_stream_close(s);
              reads_done += my_reads;
              calls_done += my_calls;
              hits_done += my_hits;
            });
          while (go.load() < (int)T && !bad.load()) std::this_thread::yield();
          const auto t0 = std::chrono::steady_clock::now();
          std::this_thread::sleep_for(std::chrono::duration<double>(secs));
          stop = true;
          for (auto& x : th) x.join();
          // (the calls in flight when the clock stops run to their end: the interval is taken after the joins)
          const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          Cell c;
          c.entry = entry; c.mem = mem; c.pack_pairs = pp; c.threads = (int)T;
          c.calls = calls_done.load(); c.hits = hits_done.load();
          c.reads_per_s = reads_done.load() / dt;
          c.us_per_call = c.calls ? 1e6 * dt * (double)T / (double)c.calls : 0.0;
          c.ok = bad.load() == 0;
          cells.push_back(c);
          fprintf(stderr, "%-6s %-8s pack %8lld pairs x %2d threads: %9.2f M reads/s  %9.1f us/call  %s%s\n", entry.c_str(), mem.c_str(), pp,
                  (int)T, c.reads_per_s / 1e6, c.us_per_call, c.ok ? "ok" : "FAILED: ", c.ok ? "" : err.c_str());
          if (!c.ok) { fprintf(stderr, "stopping\n"); return 1; }
        }
      }
  // latency of one-read calls (Indexer::map_read as the reference calls it)
  gf_seqmatch one[2];
  for (int r = 0; r < 300; ++r) gf_map_read(ix, pageable + offsets[(size_t)r], offsets[(size_t)r + 1] - offsets[(size_t)r], one);
  const auto a = std::chrono::steady_clock::now();
  const int N1 = 3000;
  for (int r = 0; r < N1; ++r) gf_map_read(ix, pageable + offsets[(size_t)r], offsets[(size_t)r + 1] - offsets[(size_t)r], one);
  const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / N1;

  printf("{\"n_reads\": %lld, \"seconds_per_cell\": %.2f, \"checked\": %s, \"us_per_single_read_call\": %.2f, \"cells\": [", (long long)n, secs,
         check ? "true" : "false", us1);
  for (size_t i = 0; i < cells.size(); ++i) {
    const Cell& c = cells[i];
    printf("%s{\"entry\": \"%s\", \"mem\": \"%s\", \"pack_pairs\": %lld, \"threads\": %d, \"reads_per_s\": %.0f, \"us_per_call\": %.1f, "
           "\"calls\": %lld, \"hits\": %lld}", i ? ", " : "", c.entry == "hits" ? "gf_map_reads_hits" : "gf_stream_submit/collect (depth 3)",
           c.mem.c_str(), c.pack_pairs, c.threads, c.reads_per_s, c.us_per_call, c.calls, c.hits);
  }
  printf("]}\n");
  gf_host_free(pinned);
  gf_host_free(pinned_off);
  gf_index_free(ix);
  free(pageable);
  return 0;
}
