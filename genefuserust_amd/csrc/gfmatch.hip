// gfmatch — host side of the C ABI (include/gfmatch.h) for gfx950.
//
// No CPU fallback lives here: every compute entry point launches the HIP
// kernels of gf_index_kernels.h / gf_map_kernels.h / gf_compact_kernels.h or
// fails with an error code.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include <functional>
#include <unistd.h>

#include "../../include/gfmatch.h"
#include "gf_compact_kernels.h"
#include "gf_exchange_kernels.h"
#include "gf_index_kernels.h"
#include "gf_map_kernels.h"
#include "gf_merge_kernels.h"
#include "gf_fastq_kernels.h"
#include "gf_host_pack.h"
#include "gf_pipe_kernels.h"
#include "gf_pair_kernels.h"
#include "gf_tail_kernels.h"
#include "gf_table.h"

static_assert(sizeof(gf_seqmatch) == 16, "gf_seqmatch layout");
static_assert(sizeof(gf_hit) == 48, "gf_hit layout");
static_assert(sizeof(gf_pair_hit) == 64, "gf_pair_hit layout");
static_assert(GF_LIN_PAD >= GF_MAX_READ_LEN, "site-code padding must cover the longest read");

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define GF_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      char buf_[512];                                                                      \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
               __FILE__, __LINE__);                                                        \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? GF_ERR_NO_DEVICE \
                                                                        : GF_ERR_HIP,      \
                  buf_);                                                                   \
    }                                                                                      \
  } while (0)

// RAII: make the index's device current for the duration of a call.
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t count) { return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)); }
  T* release() {
    T* q = p;
    p = nullptr;
    return q;
  }
};

// A few host threads that stay: gf_index_build's gather of the gene slices is 0.2 ms of copying, and starting
// eight std::threads for it took 0.23 ms (r03, measured).  The pool is the process's, made on first use and never
// taken down (its threads sleep on a condition variable; a process that forked gets a new one on its first job);
// run() hands `f` to `n` of them and returns at once, wait() returns when they are all through with it.
struct HostPool {
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void()> job;
  uint64_t gen = 0;
  int wanted = 0, taken = 0, running = 0;
  int n_threads = 0;
  pid_t owner = 0;
  static HostPool& get() {
    static std::mutex make_mu;
    static HostPool* pool = nullptr;
    std::lock_guard<std::mutex> g(make_mu);
    if (!pool || pool->owner != getpid()) {
      pool = new HostPool();  // (never deleted: its threads may be asleep in it when the process ends)
      pool->owner = getpid();
    }
    return *pool;
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_go.wait(lk, [&] { return gen != seen && taken < wanted; });
      seen = gen;
      ++taken;
      std::function<void()> f = job;
      lk.unlock();
      f();
      lk.lock();
      if (--running == 0) cv_done.notify_all();
    }
  }
  // one job at a time (gf_index_build holds the staging block's lock); returns how many threads took it on —
  // fewer than asked for on a host that will not start another thread, 0: do it yourself, nothing to wait() for
  int run(int n, std::function<void()> f) {
    std::unique_lock<std::mutex> lk(mu);
    while (n_threads < n) {
      try {
        std::thread(&HostPool::worker, this).detach();
      } catch (...) {
        break;
      }
      ++n_threads;
    }
    n = std::min(n, n_threads);
    if (n <= 0) return 0;
    job = std::move(f);
    wanted = running = n;
    taken = 0;
    ++gen;
    lk.unlock();
    cv_go.notify_all();
    return n;
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return running == 0; });
    job = nullptr;
  }
};

// Device blocks of freed indexes, kept for the next gf_index_build on the same device: in multi-CSV mode an
// index lives for one pass over the reads (fusion_scan.rs:62-188), and hipMalloc + hipFree of its arrays cost
// 1.3 ms per rebuild next to 2-6 ms of build kernels.  A request takes the smallest cached block that is large
// enough and at most a quarter larger; at most GF_INDEX_CACHE_MIB (default 2048) stay cached per process,
// gf_index_trim frees them.
struct BlockCache {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> idle;  // (device, bytes) -> block
  std::map<void*, size_t> size_of;                    // every block handed out or idle -> its size
  size_t idle_bytes = 0;
};
BlockCache& block_cache() {
  static BlockCache* c = new BlockCache();  // (never destroyed: no hipFree after the runtime has shut down)
  return *c;
}
size_t block_cache_limit() {
  static const size_t lim = [] {
    const char* e = getenv("GF_INDEX_CACHE_MIB");
    return (size_t)(e ? std::max(0l, atol(e)) : 2048l) << 20;
  }();
  return lim;
}
hipError_t block_alloc(int dev, void** out, size_t bytes) {
  bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
  BlockCache& C = block_cache();
  {
    std::lock_guard<std::mutex> g(C.mu);
    auto it = C.idle.lower_bound({dev, bytes});
    if (it != C.idle.end() && it->first.first == dev && it->first.second <= bytes + bytes / 4 + (64u << 10)) {
      *out = it->second;
      C.idle_bytes -= it->first.second;
      C.idle.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {  // give the idle blocks back and try once more
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(C.mu);
      for (auto it = C.idle.begin(); it != C.idle.end();)
        if (it->first.first == dev) {
          drop.push_back(it->second);
          C.idle_bytes -= it->first.second;
          C.size_of.erase(it->second);
          it = C.idle.erase(it);
        } else {
          ++it;
        }
    }
    for (void* q : drop) (void)hipFree(q);
    (void)hipGetLastError();
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(C.mu);
  C.size_of[*out] = bytes;
  return hipSuccess;
}
// (the caller has waited for the device: nothing in flight reads the block any more)
void block_free(int dev, void* p) {
  if (!p) return;
  BlockCache& C = block_cache();
  {
    std::lock_guard<std::mutex> g(C.mu);
    auto it = C.size_of.find(p);
    if (it != C.size_of.end() && C.idle_bytes + it->second <= block_cache_limit()) {
      C.idle.insert({{dev, it->second}, p});
      C.idle_bytes += it->second;
      return;
    }
    if (it != C.size_of.end()) C.size_of.erase(it);
  }
  (void)hipFree(p);
}
void block_trim(int dev) {
  BlockCache& C = block_cache();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(C.mu);
    for (auto it = C.idle.begin(); it != C.idle.end();)
      if (it->first.first == dev) {
        drop.push_back(it->second);
        C.idle_bytes -= it->first.second;
        C.size_of.erase(it->second);
        it = C.idle.erase(it);
      } else {
        ++it;
      }
  }
  for (void* q : drop) (void)hipFree(q);
}

}  // namespace

struct gf_index {
  int device = 0;
  int n_cus = 256;
  GfTable table{};
  uint64_t* d_slots = nullptr;
  uint32_t* d_dupes = nullptr;
  uint32_t* d_lin_base = nullptr;
  uint32_t* d_lin_hi = nullptr;
  uint32_t* d_gene_len = nullptr;
  uint32_t* d_gdu = nullptr;
  uint32_t* d_gdt = nullptr;  // the same in overlapping 128-byte tiles (GfTable::gdt)
  uint32_t* d_bloom = nullptr;
  // first pass for reads <= 256 bases: 0 = flat pipeline (pack, seed+verify, probe, exact
  // kernel on survivors), 1 = wave-per-read probe-all, 2 = wave-per-read seed+verify
  int map_variant = 0;
  int64_t pack_call_reads = -1;  // host calls of up to this many reads take the zero-copy route (-1: GF_PACK_CALL_READS / the default)
  // Indexer.m_fusion_seq (indexer.rs:77, :170): the upper-cased gene slices.  The device keeps them (d_cat, one
  // concatenation: the build's own input, upper-cased there); the host copy of a gene is fetched on first use —
  // in multi-CSV mode an index lives for one pass over the reads and only the genes of its few hits are ever asked for
  mutable std::vector<std::string> fusion_seq;
  mutable std::vector<uint8_t> fusion_ready;
  mutable std::mutex fusion_mu;
  std::vector<uint32_t> gene_off_h, gene_len_h;
  uint8_t* d_cat = nullptr;
  uint32_t* d_gene_off = nullptr;
  int ensure_fusion(int32_t c) const;
  gf_index_info info{};
  // profiling
  bool profiling = false;
  bool have_events = false;
  bool recorded = false;
  uint8_t* d_gene_rev = nullptr;  // Fusion::is_reversed() per gene (gf_index_set_gene_reversed)
  std::atomic<int> open_streams{0};  // gf_streams opened on this index and not closed yet
  // lanes of the host-buffer entry points (gf_map_reads, gf_map_read, gf_map_reads_hits): see HostLane
  std::vector<struct HostLane*> lanes;
  std::mutex lane_mu;
  std::condition_variable lane_cv;
  hipEvent_t ev0{}, ev1{};
  hipEvent_t ev_stage[5]{};  // pipeline stage boundaries: seed+verify | filter | buckets | exact kernel
  bool stages_recorded = false;
  std::mutex prof_mu;

  void free_lanes();
  ~gf_index() {
    // hipFree used to wait for the device; the blocks go back to the cache instead, so wait here: work queued
    // by the asynchronous entry points may still read the table
    (void)hipDeviceSynchronize();
    block_free(device, d_slots);
    block_free(device, d_dupes);
    block_free(device, d_lin_base);
    block_free(device, d_lin_hi);
    block_free(device, d_gene_len);
    block_free(device, d_gdu);
    block_free(device, d_gdt);
    block_free(device, d_bloom);
    block_free(device, d_cat);
    block_free(device, d_gene_off);
    if (d_gene_rev) (void)hipFree(d_gene_rev);
    free_lanes();
    if (have_events) {
      (void)hipEventDestroy(ev0);
      (void)hipEventDestroy(ev1);
      for (auto& e : ev_stage) (void)hipEventDestroy(e);
    }
  }
};

// Workspaces of the device-buffer entry points: one per (device, stream, kind), grow-only, owned
// by the PROCESS, not by an index — in multi-CSV mode the index is rebuilt per CSV over a resident
// read set (fusion_scan.rs:62-188), and a workspace that died with its index meant a multi-GB
// hipFree + hipMalloc per CSV.  Calls on one stream run in order, so the next call may reuse the
// buffers; launch sequences are serialised by the kind's mutex so that two host threads sharing a
// stream cannot interleave their kernels.  (Stream-ordered hipMallocAsync/hipFreeAsync was tried
// first: on the legacy default stream a free issued right after the launches raced with the kernels.)
// gf_index_trim releases a device's workspaces; nothing is freed at process exit.
struct Workspace {
  void* base = nullptr;
  size_t bytes = 0;
  std::mutex mu;  // this entry's buffers and the launch sequences that use them (one stream's calls stay in order)
};
struct WsKey {
  int device; hipStream_t st;
  bool operator<(const WsKey& o) const { return device != o.device ? device < o.device : st < o.st; }
};
// The pool's own mutex guards the map only; growing a workspace (stream sync, hipFree, hipMalloc) and queueing
// a call's launches happen under the ENTRY's mutex, so callers on other streams and other GPUs are not held up
// (one pool-wide mutex stalled every caller behind one thread's hipMalloc).
struct WsPool {
  std::mutex mu;
  std::map<WsKey, std::shared_ptr<Workspace>> ws;
  std::shared_ptr<Workspace> entry(int device, hipStream_t st) {
    std::lock_guard<std::mutex> g(mu);
    auto& e = ws[WsKey{device, st}];
    if (!e) e = std::make_shared<Workspace>();
    return e;
  }
  // forget (and free) the workspace of a stream that is about to be destroyed; the caller has drained the stream
  void drop(int device, hipStream_t st) {
    std::shared_ptr<Workspace> e;
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = ws.find(WsKey{device, st});
      if (it == ws.end()) return;
      e = it->second;
      ws.erase(it);
    }
    std::lock_guard<std::mutex> g(e->mu);
    if (e->base) (void)hipFree(e->base);
    e->base = nullptr;
    e->bytes = 0;
  }
};
static WsPool& ws_pool(bool pair) {
  static WsPool* pools = new WsPool[2];  // (never destroyed: the HIP runtime may be gone by then)
  return pools[pair ? 1 : 0];
}

// one launch of gf_k_compact_scan: one array, or two of the same length (block b scans array b)
static void launch_scan(hipStream_t st, int64_t ntiles, const uint32_t* c0, int64_t* o0, int64_t* t0,
                        const uint32_t* c1 = nullptr, int64_t* o1 = nullptr, int64_t* t1 = nullptr) {
  GfScanJobs J;
  J.j[0] = GfScanJob{c0, o0, t0};
  J.j[1] = GfScanJob{c1, o1, t1};
  hipLaunchKernelGGL(gf_k_compact_scan, dim3(c1 ? 2 : 1), dim3(1024), 0, st, J, ntiles);
}

// the same for very many totals, with scratch for the rounds' sums (uint32 per round) and bases (int64 per round):
// every total must be small enough for a round's 16 384 of them to sum below 2^32
static inline int64_t scan_rounds(int64_t ntiles) { return (ntiles + GF_SCAN_ROUND - 1) / GF_SCAN_ROUND; }
static void launch_scan_big(hipStream_t st, int64_t ntiles, const uint32_t* c0, int64_t* o0, int64_t* t0, uint32_t* round_sums,
                            int64_t* round_offsets) {
  const int64_t nr = scan_rounds(ntiles);
  if (nr <= 2) {
    launch_scan(st, ntiles, c0, o0, t0);
    return;
  }
  hipLaunchKernelGGL(gf_k_compact_scan_rounds, dim3((unsigned)nr), dim3(1024), 0, st, c0, ntiles, o0, round_sums);
  launch_scan(st, nr, round_sums, round_offsets, t0);
  hipLaunchKernelGGL(gf_k_compact_scan_add, dim3((unsigned)nr), dim3(1024), 0, st, o0, ntiles, (const int64_t*)round_offsets);
}

static int acquire_workspace(Workspace& W, hipStream_t st, size_t need, void** out) {  // caller holds W.mu
  if (W.bytes < need) {
    if (W.base) {
      GF_HIP(hipStreamSynchronize(st));  // earlier calls on this stream may still use it
      GF_HIP(hipFree(W.base));
      W.base = nullptr;
      W.bytes = 0;
    }
    // a quarter more than asked for: packs of slightly different sizes (a streamed FASTQ) must not
    // free and allocate gigabytes every time one is a little larger than the last
    const size_t want = need + need / 4;
    if (hipMalloc(&W.base, want) == hipSuccess) {
      W.bytes = want;
    } else {
      (void)hipGetLastError();
      GF_HIP(hipMalloc(&W.base, need));
      W.bytes = need;
    }
  }
  *out = W.base;
  return GF_OK;
}

// ---- the flat pipeline's four launches for one span of reads ----
struct FlatWs {
  void* list_b;
  uint32_t* list_c;
  uint32_t* list_long;
  unsigned int* blk_cnt;
  unsigned int* blk_cnt2;
  unsigned int* ctr;
};
struct FlatPlan {
  int nblk;
  int64_t per_block;
  size_t sz_lb, sz_lc, sz_ll, sz_bc, sz_ctr;
  size_t bytes() const { return sz_lb + sz_lc + sz_ll + 2 * sz_bc + sz_ctr; }
};
// pw = words per read of the flat kernels (10, 16 or 20); long_reads = the batch may hold reads
// beyond the flat kernels' limit
static FlatPlan flat_plan(int64_t n, int n_cus, int pw, bool long_reads) {
  FlatPlan p;
  // Up to 32 blocks per CU, but about 3 K reads a block at least and whole rounds of the four blocks a CU runs at a time:
  // 20 M reads in 6144 blocks were 1.4 % faster than in 8192 (the list kernels pay per block: 0.155 -> 0.133 ms),
  // 200 M reads in 8192 blocks 0.4 % faster than in 6144 (r03 b; GF_NBLK_MULT sets blocks per CU outright: experiments)
  const int64_t round_blocks = (int64_t)n_cus * 4;
  int64_t nblk = std::min<int64_t>((int64_t)n_cus * 32, (n / 3072 + round_blocks / 2) / round_blocks * round_blocks);  // (the nearest whole round)
  nblk = std::max<int64_t>(nblk, round_blocks);
  if (const char* e = getenv("GF_NBLK_MULT")) nblk = (int64_t)n_cus * std::max(1, atoi(e));
  p.nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, nblk));
  p.per_block = (n + p.nblk - 1) / p.nblk;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t esz = pw == 10 ? sizeof(GfPipeEntryW<10>) : (pw == 16 ? sizeof(GfPipeEntryW<16>) : sizeof(GfPipeEntryW<20>));
  p.sz_lb = al((size_t)n * esz);
  p.sz_lc = al((size_t)n * sizeof(uint32_t));
  p.sz_ll = long_reads ? al((size_t)n * sizeof(uint32_t)) : 0;
  p.sz_bc = al((size_t)p.nblk * sizeof(unsigned int));
  p.sz_ctr = 256;
  return p;
}
static FlatWs flat_carve(uint8_t* wp, const FlatPlan& p) {
  FlatWs w;
  w.list_b = (void*)wp; wp += p.sz_lb;
  w.list_c = (uint32_t*)wp; wp += p.sz_lc;
  w.list_long = (uint32_t*)wp; wp += p.sz_ll;
  w.blk_cnt = (unsigned int*)wp; wp += p.sz_bc;
  w.blk_cnt2 = (unsigned int*)wp; wp += p.sz_bc;
  w.ctr = (unsigned int*)wp;
  return w;
}
// ev[0..4] (optional): stage boundaries.  lmax = longest read of the flat kernels in this call,
// batch_max = the caller's limit: reads in between go to the wave-per-read kernels of the
// longer classes (LDS footprints for 1024 and 4096 bases) through a list.
// the reads of a call: ASCII bases, or their packed form (gf_pack_bases_device); offsets count bases either way
struct ReadSrc {
  const uint8_t* bases = nullptr;
  const uint32_t* pk = nullptr;
  const uint16_t* iv = nullptr;
  bool packed() const { return pk != nullptr; }
};

template <int PW, bool PACKED>
static int launch_flat(const gf_index* idx, const GfTable& T, hipStream_t st, const ReadSrc& src, const int64_t* offsets,
                       int64_t n, int lmax, int batch_max, uint8_t* counts, gf_seqmatch* matches, const FlatWs& w,
                       const FlatPlan& p, hipEvent_t* ev) {
  const uint8_t* bases = src.bases;
  GF_HIP(hipMemsetAsync(w.ctr, 0, 64, st));
  if (ev) GF_HIP(hipEventRecord(ev[0], st));
  // Four blocks per CU: r03 padded the PW = 10 kernel's LDS to get there (2.5 % faster than six); since r04 its queue
  // takes 24 KB itself — 40 KB per block, four per CU without padding (three cost 5.7 %, five or a smaller queue change
  // nothing: the kernel is bound by instruction issue).  GF_SV_INLINE_ROUNDS builds, the r03 form, pad as before.
#ifdef GF_SV_INLINE_ROUNDS
  size_t pad_lds = PW == 10 ? 24000 : 0;
#else
  size_t pad_lds = 0;
#endif
  if (const char* e = getenv("GF_SV_PAD_LDS")) pad_lds = (size_t)atoi(e);  // experiments
  hipLaunchKernelGGL((gf_k_seedverify_stream<PW, PACKED>), dim3(p.nblk), dim3(256), pad_lds, st, T, bases, src.pk, src.iv, offsets, n,
                     lmax, batch_max, counts, (GfPipeEntryW<PW>*)w.list_b, w.blk_cnt, p.per_block, w.list_long, w.ctr);
  if (ev) GF_HIP(hipEventRecord(ev[1], st));
  // a filter that does not fit an XCD's L2 (bloom_in_l2 == 1) is asked part by part, so that the part
  // in use stays there (gf_k_probe_filter); GF_FILTER_PARTS sets the number of parts (experiments)
  static const int parts_env = getenv("GF_FILTER_PARTS") ? atoi(getenv("GF_FILTER_PARTS")) : 0;
  int nparts = T.bloom_in_l2 == 1 ? 2 : 1;
  if (parts_env >= 1 && parts_env <= 8 && T.bloom_in_l2 == 1) nparts = parts_env;
  unsigned int *cnt_in = w.blk_cnt, *cnt_out = w.blk_cnt2;
  for (int ph = 0; ph < nparts; ++ph) {
    hipLaunchKernelGGL((gf_k_probe_filter<PW>), dim3(p.nblk), dim3(256), 0, st, T,
                       (GfPipeEntryW<PW>*)w.list_b, (const unsigned int*)cnt_in, p.per_block, counts, cnt_out, ph,
                       nparts);
    std::swap(cnt_in, cnt_out);
  }
  const unsigned int* survivors = cnt_in;  // (the last launch's output)
  if (ev) GF_HIP(hipEventRecord(ev[2], st));
  hipLaunchKernelGGL((gf_k_probe_buckets<PW>), dim3(p.nblk), dim3(256), 0, st, T,
                     (const GfPipeEntryW<PW>*)w.list_b, survivors, p.per_block, counts, w.list_c, w.ctr);
  if (ev) GF_HIP(hipEventRecord(ev[3], st));
  if (PW <= 16)
    hipLaunchKernelGGL((gf_k_map_reads_list<256, 4, PACKED>), dim3(idx->n_cus * 8), dim3(256), 0, st, T, bases, src.pk, src.iv, offsets,
                       w.list_c, (int64_t)1, w.ctr + 1, counts, matches);
  else  // survivors of up to 320 bases
    hipLaunchKernelGGL((gf_k_map_reads_list<1024, 4, PACKED>), dim3(idx->n_cus * 4), dim3(256), 0, st, T, bases, src.pk, src.iv,
                       offsets, w.list_c, (int64_t)1, w.ctr + 1, counts, matches);
  if (batch_max > lmax) {
    hipLaunchKernelGGL((gf_k_map_reads_list<1024, 4, PACKED>), dim3(idx->n_cus * 4), dim3(256), 0, st, T, bases, src.pk, src.iv,
                       offsets, w.list_long, (int64_t)1, w.ctr + 2, counts, matches);
    if (batch_max > 1024)
      hipLaunchKernelGGL((gf_k_map_reads_list<4096, 2, PACKED>), dim3(idx->n_cus * 4), dim3(128), 0, st, T, bases, src.pk, src.iv,
                         offsets, w.list_long + (n - 1), (int64_t)-1, w.ctr + 3, counts, matches);
  }
  GF_HIP(hipGetLastError());
  if (ev) GF_HIP(hipEventRecord(ev[4], st));
  return GF_OK;
}

extern "C" {

const char* gf_last_error(void) { return g_err.c_str(); }
const char* gf_version(void) { return "gfmatch 0.4.0 (gfx950)"; }

int gf_index_build(const char* const* gene_seqs, const int64_t* gene_lens, int32_t n_genes,
                   const gf_options* opts, gf_index** out_index) {
  if (!out_index) return fail(GF_ERR_ARG, "out_index is null");
  *out_index = nullptr;
  if (n_genes < 0 || n_genes > 32767) return fail(GF_ERR_ARG, "n_genes must be in 0..32767 (contig is i16)");
  if (n_genes > 0 && (!gene_seqs || !gene_lens)) return fail(GF_ERR_ARG, "gene arrays are null");

  int dev = opts ? opts->device : -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(GF_ERR_NO_DEVICE, "no HIP device available (gfmatch has no CPU fallback)");
  if (dev < 0) GF_HIP(hipGetDevice(&dev));
  if (dev >= ndev) return fail(GF_ERR_NO_DEVICE, "device ordinal out of range");
  DeviceGuard guard(dev);
  if (!guard.ok) return fail(GF_ERR_NO_DEVICE, "hipSetDevice failed");

  // GF_BUILD_TIMING=1: phase times of this call on stderr (experiments: multi-CSV mode rebuilds per CSV)
  static const bool timing = getenv("GF_BUILD_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_start = now();
  auto t_entry = now();
  auto mark = [&](const char* what) {  // host clock only, no synchronisation
    if (!timing) return;
    fprintf(stderr, "[gf_index_build]   . %-40s at %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now() - t_entry).count());
  };
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    auto t = now();
    fprintf(stderr, "[gf_index_build] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_start).count());
    t_start = t;
  };
  std::unique_ptr<gf_index> ix(new gf_index());
  ix->device = dev;
  if (const char* e = getenv("GF_MAP_VARIANT")) ix->map_variant = atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : 0;  // experiments
  {
    static std::atomic<int> cus_of[64];  // (hipGetDeviceProperties is 0.1 ms a call: once per device and process)
    int cus = dev < 64 ? cus_of[dev].load() : 0;
    if (cus <= 0) {
      hipDeviceProp_t prop;
      GF_HIP(hipGetDeviceProperties(&prop, dev));
      cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      if (dev < 64) cus_of[dev].store(cus);
    }
    ix->n_cus = cus;
  }

  // --- host: upper-case the slices (indexer.rs:159), lay out the site-code space ---
  std::vector<uint32_t> gene_off((size_t)n_genes + 1, 0), lin_base((size_t)std::max(n_genes, 1), 0),
      lin_hi((size_t)std::max(n_genes, 1), 0), glen((size_t)std::max(n_genes, 1), 0);
  uint64_t total = 0, lin_cursor = 0, site_bound = 0;
  ix->fusion_seq.resize((size_t)n_genes);
  for (int32_t c = 0; c < n_genes; ++c) {
    int64_t len = gene_lens[c] < 0 ? 0 : gene_lens[c];
    if (len > 0 && !gene_seqs[c]) return fail(GF_ERR_ARG, "gene sequence pointer is null");
    if (len > (int64_t)GF_LIN_MASK) return fail(GF_ERR_CAPACITY, "gene longer than the site-code space");
    gene_off[(size_t)c] = (uint32_t)total;
    total += (uint64_t)len;
    lin_base[(size_t)c] = (uint32_t)(lin_cursor + (uint64_t)len + GF_LIN_PAD);
    lin_cursor += 2ull * (uint64_t)len + GF_LIN_PAD;
    lin_hi[(size_t)c] = (uint32_t)lin_cursor;
    glen[(size_t)c] = (uint32_t)len;
    if (len > GF_KMER) site_bound += 2ull * (uint64_t)(len - GF_KMER);
    if (lin_cursor > (uint64_t)GF_LIN_MASK || total > 0xFFFFFFFFull - 2 * GF_TILE_BASES)
      return fail(GF_ERR_CAPACITY, "gene set too large: total span exceeds the 29-bit site-code space");
  }
  gene_off[(size_t)n_genes] = (uint32_t)total;

  ix->fusion_ready.assign((size_t)n_genes, 0);
  ix->gene_off_h = gene_off;
  ix->gene_len_h.assign(glen.begin(), glen.begin() + n_genes);

  // 8 slots per bucket, about 4 keys per bucket on average
  uint64_t sites_per_bucket = 4;
  if (const char* e = getenv("GF_SITES_PER_BUCKET")) sites_per_bucket = (uint64_t)std::max(1, std::min(7, atoi(e)));  // experiments
  uint64_t nb64 = std::max<uint64_t>(16, (site_bound + sites_per_bucket - 1) / sites_per_bucket);
  if (nb64 > 0x7FFFFFFFull) return fail(GF_ERR_CAPACITY, "table too large");
  const uint32_t nbuckets = (uint32_t)nb64;
  const uint64_t nslots = (uint64_t)nbuckets * GF_SLOTS_PER_BUCKET;
  const uint32_t ntiles = (uint32_t)((total + GF_TILE_BASES - 1) / GF_TILE_BASES);
  const size_t cat_bytes = (size_t)ntiles * GF_TILE_BASES + 64;

  // Device blocks first, and their clears queued at once: the device does them while the host gathers the slices.
  struct { uint8_t*& p; } d_cat{ix->d_cat};
  struct { uint32_t*& p; } d_goff{ix->d_gene_off};
  struct StatBlock {  // the statistics + the fill counts of the filter's partitions (32 bits each)
    int dev; unsigned long long* p = nullptr;
    ~StatBlock() { if (p) { (void)hipDeviceSynchronize(); block_free(dev, p); } }
  } d_stats{dev};
  const size_t stat_words = 8 + GF_FPARTS_MAX / 2;
  GF_HIP(block_alloc(dev, (void**)&ix->d_cat, cat_bytes));
  GF_HIP(block_alloc(dev, (void**)&ix->d_gene_off, ((size_t)n_genes + 1) * sizeof(uint32_t)));
  GF_HIP(block_alloc(dev, (void**)&d_stats.p, stat_words * sizeof(unsigned long long)));
  // both strands of the genes in site-code space + per-site uniqueness bits (diagonal
  // verification of the mapping kernel); padded so that a 256-base read hanging over
  // either end of the space stays inside the arrays
  const size_t gd_words = (size_t)(lin_cursor / 16) + 64;
  GF_HIP(block_alloc(dev, (void**)&ix->d_gdu, 2 * gd_words * sizeof(uint32_t)));
  GF_HIP(block_alloc(dev, (void**)&ix->d_slots, nslots * sizeof(uint64_t)));
  GF_HIP(block_alloc(dev, (void**)&ix->d_lin_base, lin_base.size() * sizeof(uint32_t)));
  GF_HIP(block_alloc(dev, (void**)&ix->d_lin_hi, lin_hi.size() * sizeof(uint32_t)));
  GF_HIP(block_alloc(dev, (void**)&ix->d_gene_len, glen.size() * sizeof(uint32_t)));
  GF_HIP(hipMemsetAsync(ix->d_gdu, 0, 2 * gd_words * sizeof(uint32_t), 0));
  GF_HIP(hipMemsetAsync(ix->d_slots, 0, nslots * sizeof(uint64_t), 0));
  GF_HIP(hipMemsetAsync(d_stats.p, 0, stat_words * sizeof(unsigned long long), 0));

  mark("blocks taken, clears queued");

  // The raw slices and the four small arrays go into ONE pinned staging block (kept by the process, grow-only) and
  // from there to the device, where the bytes are upper-cased in place (indexer.rs:159).  The staging block is the
  // process's: builds take turns at it.  (r03: the host third of a rebuild — upper-casing two copies of every gene,
  // a pageable 15 MB copy, the clears behind them — was 0.96 of 3.7 ms per cancer-shaped gene set.)
  static std::mutex stage_mu;
  static uint8_t* stage = nullptr;
  static size_t stage_bytes = 0;
  std::unique_lock<std::mutex> stage_lock(stage_mu);
  // copies out of the block are done before the next build may write it: on the early exits by this guard, on the
  // way through by the synchronous copy of the statistics below, after which the block is handed on (it
  // used to stay locked until the build returned, so builds on different devices or threads took turns for all of it)
  struct StageDrain { bool armed = true; ~StageDrain() { if (armed) (void)hipDeviceSynchronize(); } } stage_drain;
  const size_t meta_words = ((size_t)n_genes + 1) + 3 * lin_base.size();
  const size_t need_stage = cat_bytes + meta_words * sizeof(uint32_t) + 64;
  if (stage_bytes < need_stage) {
    if (stage) (void)hipHostFree(stage);
    stage = nullptr;
    stage_bytes = 0;
    GF_HIP(hipHostMalloc((void**)&stage, need_stage + need_stage / 4, hipHostMallocDefault));
    stage_bytes = need_stage + need_stage / 4;
  }
  // The gather goes megabyte by megabyte of the concatenation, a few threads of the host pool taking them in
  // order, and a megabyte's way to the device (a kernel that pulls it over the bus and upper-cases it) is queued as
  // soon as it is there.
  const uint64_t CH = 1ull << 20;
  const int nch = (int)((cat_bytes + CH - 1) / CH);
  auto gather = [&](int k) {
    const uint64_t b0 = (uint64_t)k * CH, b1 = std::min<uint64_t>(b0 + CH, cat_bytes);
    if (b0 < total) {
      // first gene that reaches into the chunk: last c with gene_off[c] <= b0
      int32_t c = (int32_t)(std::upper_bound(gene_off.begin(), gene_off.begin() + n_genes, (uint32_t)b0) - gene_off.begin()) - 1;
      for (c = std::max(c, 0); c < n_genes && gene_off[(size_t)c] < b1; ++c) {
        const uint64_t g0 = gene_off[(size_t)c], g1 = g0 + glen[(size_t)c];
        const uint64_t x0 = std::max(g0, b0), x1 = std::min(g1, b1);
        if (x1 > x0) memcpy(stage + x0, gene_seqs[c] + (x0 - g0), (size_t)(x1 - x0));
      }
    }
    if (b1 > total) memset(stage + std::max<uint64_t>(b0, total), 0, (size_t)(b1 - std::max<uint64_t>(b0, total)));
  };
  const int T = std::min(8, nch / 2);  // a thread per two megabytes, at most 8; none below four
  std::atomic<int> next_chunk{0};
  std::unique_ptr<std::atomic<int>[]> chunk_done(new std::atomic<int>[(size_t)nch]);
  for (int k = 0; k < nch; ++k) chunk_done[(size_t)k].store(0, std::memory_order_relaxed);
  struct PoolWait {  // (every return below, the failing ones too, waits for the threads: they use this frame)
    HostPool* pool = nullptr;
    ~PoolWait() { if (pool) pool->wait(); }
  } pool_wait;
  bool pooled = false;
  if (T > 0) {
    try {  // (a host that cannot start another thread gathers on this one)
      HostPool& hp = HostPool::get();
      pooled = hp.run(T, [&] {
        for (int k = next_chunk.fetch_add(1); k < nch; k = next_chunk.fetch_add(1)) {
          gather(k);
          chunk_done[(size_t)k].store(1, std::memory_order_release);
        }
      }) > 0;
      if (pooled) pool_wait.pool = &hp;
    } catch (...) {  // (bad_alloc from the job's std::function, at worst: nothing was started)
      pooled = false;
    }
  }
  mark("gather threads started");
  static const bool pull = getenv("GF_BUILD_DMA") == nullptr;  // (GF_BUILD_DMA=1: copies by the DMA engine + gf_k_upper_inplace, as before)
  for (int k = 0; k < nch; ++k) {
    if (!pooled) gather(k);
    else while (!chunk_done[(size_t)k].load(std::memory_order_acquire)) std::this_thread::yield();
    const uint64_t b0 = (uint64_t)k * CH, b1 = std::min<uint64_t>(b0 + CH, cat_bytes);  // (both multiples of 16)
    if (pull) {
      hipLaunchKernelGGL(gf_k_upper_copy, dim3((unsigned)(((b1 - b0) / 16 + 255) / 256)), dim3(256), 0, 0,
                         (const uint8_t*)stage + b0, d_cat.p + b0, (unsigned long long)((b1 - b0) / 16));
      GF_HIP(hipGetLastError());
    } else {
      GF_HIP(hipMemcpyAsync(d_cat.p + b0, stage + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, 0));
    }
  }
  mark("last chunk queued");
  uint32_t* meta = (uint32_t*)(stage + ((cat_bytes + 63) & ~(size_t)63));
  uint32_t* m_goff = meta; uint32_t* m_lb = m_goff + ((size_t)n_genes + 1);
  uint32_t* m_lh = m_lb + lin_base.size(); uint32_t* m_gl = m_lh + lin_base.size();
  memcpy(m_goff, gene_off.data(), gene_off.size() * sizeof(uint32_t));
  memcpy(m_lb, lin_base.data(), lin_base.size() * sizeof(uint32_t));
  memcpy(m_lh, lin_hi.data(), lin_hi.size() * sizeof(uint32_t));
  memcpy(m_gl, glen.data(), glen.size() * sizeof(uint32_t));
  lap("host: slices into the pinned block and on to the device (clears queued)");
  GF_HIP(hipMemcpyAsync(d_goff.p, m_goff, gene_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, 0));
  GF_HIP(hipMemcpyAsync(ix->d_lin_base, m_lb, lin_base.size() * sizeof(uint32_t), hipMemcpyHostToDevice, 0));
  GF_HIP(hipMemcpyAsync(ix->d_lin_hi, m_lh, lin_hi.size() * sizeof(uint32_t), hipMemcpyHostToDevice, 0));
  GF_HIP(hipMemcpyAsync(ix->d_gene_len, m_gl, glen.size() * sizeof(uint32_t), hipMemcpyHostToDevice, 0));
  if (!pull) {
    hipLaunchKernelGGL(gf_k_upper_inplace, dim3((unsigned)std::min<size_t>((cat_bytes / 16 + 255) / 256, 4096)), dim3(256), 0, 0,
                       d_cat.p, (unsigned long long)(cat_bytes / 16));
    GF_HIP(hipGetLastError());
  }

  lap("alloc, memset, copies in");
  GfGenes G;
  G.cat = d_cat.p;
  G.gene_off = d_goff.p;
  G.lin_base = ix->d_lin_base;
  G.total = (uint32_t)total;
  G.n_genes = n_genes;

  const int sweep_grid = (int)std::min<uint64_t>((nslots + 255) / 256, (uint64_t)ix->n_cus * 16);
  unsigned long long stats[8] = {0};
  // both strands in site-code space: needs nothing from the table
  hipLaunchKernelGGL(gf_k_index_strands, dim3((unsigned)((gd_words + 255) / 256)), dim3(256), 0, 0, G,
                     (const uint32_t*)ix->d_lin_hi, ix->d_gdu, (uint32_t)gd_words);
  GF_HIP(hipGetLastError());
  // One pass (default): a site is written with the claim of its key's slot; the sites of keys found again go
  // through a side list (gf_index_kernels.h).  GF_BUILD_TWO_PASS=1: the COUNT / FILL pair (experiments).
  static const bool two_pass = getenv("GF_BUILD_TWO_PASS") != nullptr;
  uint32_t bloom_words = 0, bloom_in_l2 = 0;
  // presence filter over canonical 14-mers, filled with the sites.  Indexes up to ~14 M keys: <= GF_BLOOM_KIB
  // (default 3 MiB) so that it lives in every XCD's L2, and seed+verify runs the filter pass for reads
  // without a candidate diagonal itself.  Up to ~30 M keys: 2.2 bits per key, up to
  // GF_BLOOM_MID_KIB (default 8 MiB) — no longer L2-resident, but still mostly L2 hits: used for
  // the seeds and by the filter kernel, which asks it half by half so that the half in use does
  // stay in the L2 (IDX-C, 29 M keys: 5.0 G reads/s; 4.65 in one pass).  Larger indexes: about
  // GF_BLOOM_BIG_BPK (default 4) bits per key, resident in the Infinity Cache and used by the
  // filter kernel only — a lookup is then an L2-missing request like a bucket probe, but one
  // lookup answers for two windows and a negative answer spares both bucket probes.
  // (`keys`: the number of keys where it is known before the filter is filled — the two-pass build — and the
  // number of sites, 3 % more on these gene sets, where it is not.)
  auto make_filter = [&](uint64_t keys) -> int {
    size_t kib = 3072, mid_kib = 8192, big_bpk = 4;
    if (const char* e = getenv("GF_BLOOM_KIB")) kib = (size_t)atol(e);
    if (const char* e = getenv("GF_BLOOM_MID_KIB")) mid_kib = (size_t)atol(e);
    if (const char* e = getenv("GF_BLOOM_BIG_BPK")) big_bpk = (size_t)atol(e);
    const uint64_t cap_words = (uint64_t)kib * 1024 / 4, mid_words = (uint64_t)mid_kib * 1024 / 4;
    const uint64_t want_words = keys * 7 / 128;  // 1.75 bits per key: what the L2-resident form needs at least
    // beyond that: 2.2 bits per key — at 1.75 half of the background reads outlive the filter and
    // go on to the buckets (IDX-C: buckets 0.64 -> 0.37 ms, filter 1.48 -> 1.58 ms per 20 M reads)
    uint64_t mid_bpk100 = 220;
    if (const char* e = getenv("GF_BLOOM_BPK100")) mid_bpk100 = (uint64_t)atol(e);  // experiments
    const uint64_t want_mid = keys * mid_bpk100 / 3200;
    uint64_t words = std::min(std::max<uint64_t>(1024, keys / 2), cap_words);  // up to 16 bits per key
    if (kib > 0 && want_words <= cap_words) {
      bloom_in_l2 = 2;
    } else if (kib > 0 && want_mid <= mid_words) {
      words = want_mid;
      // (r03, measured on IDX-C with the vote bound's half look-ups: asking this filter from seed+verify itself,
      //  GF_BLOOM_INLINE_MID=1, 3.87 ms per 20 M reads at 1.5 bits per key, 4.13 at 2.2; the sweeps over the list
      //  3.93 at 2.2 — no winner: about half of an 8 MiB filter's look-ups miss a 4 MiB L2 either way)
      static const bool inline_mid = getenv("GF_BLOOM_INLINE_MID") && atoi(getenv("GF_BLOOM_INLINE_MID")) == 1;
      bloom_in_l2 = inline_mid ? 2 : 1;
    } else if (kib > 0 && big_bpk > 0) {
      words = std::min<uint64_t>(keys * big_bpk / 32 + 1024, (64ull << 20) / 4);
    } else {
      words = 0;
    }
    if (words) {
      bloom_words = (uint32_t)words;
      GF_HIP(block_alloc(dev, (void**)&ix->d_bloom, (size_t)bloom_words * sizeof(uint32_t)));
      GF_HIP(hipMemsetAsync(ix->d_bloom, 0, (size_t)bloom_words * sizeof(uint32_t), 0));
    }
    return GF_OK;
  };
  // the duplicate lists are sized before their keys are counted: every site could be in one (the block comes
  // from the cache of freed indexes in multi-CSV mode); the sweep that assigns them counts the keys on its way
  // (+ a granule per block of the sweep that hands the lists out: gf_k_classify_assign)
  // (granules: a third over what the small rounds really need at worst, and one unfinished granule per block)
  const uint64_t dupes_cap = std::max<uint64_t>(std::min<uint64_t>(site_bound + site_bound / 2 + (uint64_t)sweep_grid * GF_DUPE_GRANULE,
                                                                   (uint64_t)GF_DUPE_START_MASK + 1), 1);
  GF_HIP(block_alloc(dev, (void**)&ix->d_dupes, dupes_cap * sizeof(uint32_t)));
  struct SideBlock { int dev; GfSideEntry* p = nullptr; ~SideBlock() { if (p) { (void)hipDeviceSynchronize(); block_free(dev, p); } } } d_side{dev};
  const uint64_t side_cap = std::max<uint64_t>(site_bound, 1);
  uint64_t n_side = 0;
  if (!two_pass) {
    if (int rc = make_filter(site_bound)) return rc;
    GF_HIP(block_alloc(dev, (void**)&d_side.p, side_cap * sizeof(GfSideEntry)));
    // The filter first, by partitions and without global atomics (gf_index_kernels.h), in the side list's block,
    // which the insert pass only then starts to fill.  (Filters beyond 8 MiB — whole-genome-sized indexes — and
    // GF_FILTER_ATOMIC=1 keep the fill inside the insert pass.)
    static const bool filter_atomic = getenv("GF_FILTER_ATOMIC") != nullptr;
    bool filter_parted = false;
    unsigned int* d_part_fill = (unsigned int*)(d_stats.p + 8);  // (cleared with the statistics)
    if (ix->d_bloom && ntiles > 0 && !filter_atomic && bloom_words <= GF_FPARTS_MAX * GF_FSLICE_WORDS) {
      const uint32_t nparts = (bloom_words + GF_FSLICE_WORDS - 1) / GF_FSLICE_WORDS;
      const uint64_t room = side_cap * (sizeof(GfSideEntry) / sizeof(uint32_t)) / nparts;  // hashes per partition the block holds
      const uint64_t want = 2 * ((total + total / 8) / nparts) + 1024;  // twice an even share of (a little over) one hash per base
      const uint32_t part_cap = (uint32_t)std::min<uint64_t>(std::min(room, want), 0x7FFFFFFFull);
      if (part_cap > 0) {
        hipLaunchKernelGGL(gf_k_filter_scatter, dim3(ntiles), dim3(GF_INDEX_THREADS), 0, 0, G, ix->d_bloom, bloom_words, nparts,
                           (uint32_t*)d_side.p, part_cap, d_part_fill);
        GF_HIP(hipGetLastError());
        hipLaunchKernelGGL(gf_k_filter_build, dim3(nparts), dim3(1024), 0, 0, ix->d_bloom, bloom_words,
                           (const uint32_t*)d_side.p, part_cap, (const unsigned int*)d_part_fill);
        GF_HIP(hipGetLastError());
        filter_parted = true;
      }
    }
    if (ntiles > 0) {
      hipLaunchKernelGGL(gf_k_index_insert, dim3(ntiles), dim3(GF_INDEX_THREADS), 0, 0, G, ix->d_slots, nbuckets, ix->d_gdu,
                         filter_parted ? (uint32_t*)nullptr : ix->d_bloom, bloom_words, d_side.p, d_stats.p + 7,
                         (unsigned long long)side_cap);
      GF_HIP(hipGetLastError());
    }
  } else if (ntiles > 0) {
    hipLaunchKernelGGL(gf_k_index_sites<GF_MODE_COUNT>, dim3(ntiles), dim3(GF_INDEX_THREADS), 0, 0, G,
                       ix->d_slots, nbuckets, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
    GF_HIP(hipGetLastError());
  }
  // (few blocks: each ends with seven atomics on the same seven addresses, 14 ns apiece and one after the other —
  //  4096 blocks' worth was 60 us, more than a druggable-sized table's sweep itself)
  hipLaunchKernelGGL(gf_k_classify_assign, dim3(std::min(sweep_grid, ix->n_cus * 4)), dim3(256), 0, 0, ix->d_slots, nslots, d_stats.p);
  GF_HIP(hipGetLastError());
  GF_HIP(hipMemcpy(stats, d_stats.p, sizeof stats, hipMemcpyDeviceToHost));
  // Everything queued before that copy is done — the kernels and copies that read the staging block among it: the
  // block (and with it the host pool, one job at a time) goes to the next build while this one finishes its lists.
  if (pool_wait.pool) {
    pool_wait.pool->wait();
    pool_wait.pool = nullptr;
  }
  stage_drain.armed = false;
  stage_lock.unlock();
  lap(two_pass ? "strands, count pass, list assignment + statistics" : "strands, insert pass (sites, flags, filter), list assignment + statistics");
  const uint64_t dupes_extent = stats[6];  // (>= stats[5], the sites in the lists: granules end unused)
  if (dupes_extent > dupes_cap)
    return fail(GF_ERR_CAPACITY, "too many duplicated sites for the 26-bit duplicate index");
  GF_HIP(hipMemsetAsync(ix->d_dupes, 0xFF, std::max<uint64_t>(dupes_extent, 1) * sizeof(uint32_t), 0));
  if (!two_pass) {
    n_side = stats[7];
    if (n_side > side_cap) return fail(GF_ERR_CAPACITY, "side list of the index build overflowed");
    if (n_side > 0) {
      const int grid = (int)std::min<uint64_t>((n_side + 255) / 256, (uint64_t)ix->n_cus * 16);
      hipLaunchKernelGGL(gf_k_index_side, dim3(grid), dim3(256), 0, 0, (const GfSideEntry*)d_side.p, (unsigned long long)n_side,
                         ix->d_slots, nbuckets, ix->d_dupes, ix->d_gdu);
      GF_HIP(hipGetLastError());
    }
    lap("side list (duplicate lists, flags)");
  } else {
    if (int rc = make_filter(stats[1])) return rc;
    if (ntiles > 0) {
      hipLaunchKernelGGL(gf_k_index_sites<GF_MODE_FILL>, dim3(ntiles), dim3(GF_INDEX_THREADS), 0, 0, G,
                         ix->d_slots, nbuckets, ix->d_dupes, ix->d_gdu, ix->d_bloom, bloom_words);
      GF_HIP(hipGetLastError());
    }
    lap("fill pass (sites, unique flags, filter)");
  }
  if (two_pass) {  // (the one-pass build sorts a list as its last site arrives: gf_k_index_side)
    hipLaunchKernelGGL(gf_k_sort_dupes, dim3(sweep_grid), dim3(256), 0, 0, ix->d_slots, nslots, ix->d_dupes);
    GF_HIP(hipGetLastError());
  }
  // the strands + flags once more in overlapping tiles, now that the flags are final (gf_table.h: gdt)
  static const bool gdu_tiles = !(getenv("GF_GDU_TILES") && atoi(getenv("GF_GDU_TILES")) == 0);  // (0: experiments)
  if (gdu_tiles) {
    const uint32_t n_tiles = (uint32_t)((gd_words + GF_GDT_STRIDE - 1) / GF_GDT_STRIDE) + 1;
    GF_HIP(block_alloc(dev, (void**)&ix->d_gdt, (size_t)n_tiles * 128));
    hipLaunchKernelGGL(gf_k_gdu_tiles, dim3((n_tiles * 16u + 255u) / 256u), dim3(256), 0, 0, (const uint2*)ix->d_gdu,
                       (uint32_t)gd_words, (uint2*)ix->d_gdt, n_tiles);
    GF_HIP(hipGetLastError());
    ix->table.gdt = ix->d_gdt;
  }
  GF_HIP(hipDeviceSynchronize());
  lap("list sort");

  ix->table.slots = ix->d_slots;
  ix->table.bloom = ix->d_bloom;
  ix->table.bloom_words = bloom_words;
  ix->table.bloom_in_l2 = bloom_in_l2;
  ix->table.dupes = ix->d_dupes;
  ix->table.lin_base = ix->d_lin_base;
  ix->table.lin_hi = ix->d_lin_hi;
  ix->table.gene_len = ix->d_gene_len;
  ix->table.gdu = ix->d_gdu;
  ix->table.gd_words = (uint32_t)gd_words;
  ix->table.nbuckets = nbuckets;
  ix->table.n_genes = n_genes;

  gf_index_info& I = ix->info;
  I.n_genes = n_genes;
  I.total_bp = (int64_t)total;
  I.n_sites = (int64_t)stats[0];
  I.n_keys = (int64_t)stats[1];
  I.n_unique = (int64_t)stats[2];
  I.n_dupe_keys = (int64_t)stats[3];
  I.n_high_keys = (int64_t)stats[4];
  I.n_dupe_sites = (int64_t)stats[5];
  I.n_buckets = nbuckets;
  I.table_bytes = (int64_t)(nslots * sizeof(uint64_t) + dupes_cap * sizeof(uint32_t) +  // (as allocated)
                            2 * gd_words * sizeof(uint32_t) + (size_t)bloom_words * sizeof(uint32_t) +
                            (ix->d_gdt ? ((gd_words + GF_GDT_STRIDE - 1) / GF_GDT_STRIDE + 1) * 128 : 0));
  I.device = dev;
  *out_index = ix.release();
  return GF_OK;
}

void gf_index_free(gf_index* idx) {
  if (!idx) return;
  if (idx->open_streams.load() > 0) {
    // a gf_stream keeps a pointer to its index (device, table, events): freeing the index under it would be a
    // use-after-free in gf_stream_collect / gf_stream_close.  The contract (gfmatch.h) is "close the streams first";
    // a host that breaks it gets a message and a leaked index instead of a crash.
    fprintf(stderr, "gfmatch: gf_index_free with %d gf_stream(s) still open: index NOT freed (close the streams first)\n",
            idx->open_streams.load());
    g_err = "gf_index_free: gf_streams of this index are still open";
    return;
  }
  DeviceGuard guard(idx->device);
  delete idx;
}

int gf_index_info_get(const gf_index* idx, gf_index_info* out) {
  if (!idx || !out) return fail(GF_ERR_ARG, "null argument");
  *out = idx->info;
  return GF_OK;
}

// Indexer.m_fusion_seq[c] on the host: fetched from the device's upper-cased copy on first use
int gf_index::ensure_fusion(int32_t c) const {
  if (c < 0 || (size_t)c >= fusion_seq.size()) return fail(GF_ERR_ARG, "bad contig");
  std::lock_guard<std::mutex> g(fusion_mu);
  if (fusion_ready[(size_t)c]) return GF_OK;
  std::string& s = fusion_seq[(size_t)c];
  s.resize(gene_len_h[(size_t)c]);
  if (!s.empty()) {
    DeviceGuard guard(device);
    GF_HIP(hipMemcpy(&s[0], d_cat + gene_off_h[(size_t)c], s.size(), hipMemcpyDeviceToHost));
  }
  fusion_ready[(size_t)c] = 1;
  return GF_OK;
}

int64_t gf_index_fusion_seq(const gf_index* idx, int32_t contig, char* out, int64_t cap) {
  if (!idx || contig < 0 || (size_t)contig >= idx->fusion_seq.size()) return fail(GF_ERR_ARG, "bad contig");
  if (!out || cap <= 0) return (int64_t)idx->gene_len_h[(size_t)contig];  // (the length alone needs no copy)
  if (int rc = idx->ensure_fusion(contig)) return rc;
  const std::string& s = idx->fusion_seq[(size_t)contig];
  if (out && cap > 0) memcpy(out, s.data(), (size_t)std::min<int64_t>(cap, (int64_t)s.size()));
  return (int64_t)s.size();
}

int gf_index_lookup(const gf_index* idx, const uint32_t* kmers, int64_t n, int32_t* out_count,
                    int16_t* out_contig, int32_t* out_position) {
  if (!idx || n < 0 || (n > 0 && (!kmers || !out_count || !out_contig || !out_position)))
    return fail(GF_ERR_ARG, "null argument");
  if (n == 0) return GF_OK;
  DeviceGuard guard(idx->device);
  DevBuf<uint32_t> d_k;
  DevBuf<int32_t> d_c, d_p;
  DevBuf<int16_t> d_g;
  GF_HIP(d_k.alloc((size_t)n));
  GF_HIP(d_c.alloc((size_t)n));
  GF_HIP(d_g.alloc((size_t)n * 5));
  GF_HIP(d_p.alloc((size_t)n * 5));
  GF_HIP(hipMemcpy(d_k.p, kmers, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  GF_HIP(hipMemset(d_g.p, 0, (size_t)n * 5 * sizeof(int16_t)));
  GF_HIP(hipMemset(d_p.p, 0, (size_t)n * 5 * sizeof(int32_t)));
  int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)idx->n_cus * 8);
  hipLaunchKernelGGL(gf_k_lookup, dim3(grid), dim3(256), 0, 0, idx->table, d_k.p, n, d_c.p, d_g.p, d_p.p);
  GF_HIP(hipGetLastError());
  GF_HIP(hipMemcpy(out_count, d_c.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(out_contig, d_g.p, (size_t)n * 5 * sizeof(int16_t), hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(out_position, d_p.p, (size_t)n * 5 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return GF_OK;
}

// One span of a batch (n <= GF_SPAN_MAX reads): the kernels keep read indices in 32 bits.
// Test/diagnostic (like gf_index_lookup): segment_mask of the device on caller-given masks.
static int segment_mask_test_impl(const gf_index* idx, const uint8_t* masks, const int64_t* offsets, int64_t n,
                                  const int64_t* gp1, const int64_t* gp2, int32_t* out_counts, gf_seqmatch* out_matches) {
  const int64_t total = offsets[n];
  DevBuf<uint8_t> d_m, d_c;
  DevBuf<int64_t> d_o, d_g1, d_g2;
  DevBuf<gf_seqmatch> d_out;
  GF_HIP(d_m.alloc((size_t)total + 16)); GF_HIP(d_c.alloc((size_t)n)); GF_HIP(d_o.alloc((size_t)n + 1));
  GF_HIP(d_g1.alloc((size_t)n)); GF_HIP(d_g2.alloc((size_t)n)); GF_HIP(d_out.alloc(2 * (size_t)n));
  if (total > 0) GF_HIP(hipMemcpy(d_m.p, masks, (size_t)total, hipMemcpyHostToDevice));
  GF_HIP(hipMemcpy(d_o.p, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
  GF_HIP(hipMemcpy(d_g1.p, gp1, (size_t)n * 8, hipMemcpyHostToDevice));
  GF_HIP(hipMemcpy(d_g2.p, gp2, (size_t)n * 8, hipMemcpyHostToDevice));
  GF_HIP(hipMemset(d_c.p, 0, (size_t)n));
  GF_HIP(hipMemset(d_out.p, 0, 2 * (size_t)n * sizeof(gf_seqmatch)));
  const int grid = (int)std::min<int64_t>(n, (int64_t)idx->n_cus * 8);
  hipLaunchKernelGGL((gf_k_segment_mask_test<256>), dim3(grid), dim3(64), 0, 0, d_m.p, d_o.p, n, 0, d_g1.p, d_g2.p, d_c.p, d_out.p);
  hipLaunchKernelGGL((gf_k_segment_mask_test<1024>), dim3(grid), dim3(64), 0, 0, d_m.p, d_o.p, n, 256, d_g1.p, d_g2.p, d_c.p, d_out.p);
  hipLaunchKernelGGL((gf_k_segment_mask_test<4096>), dim3(grid), dim3(64), 0, 0, d_m.p, d_o.p, n, 1024, d_g1.p, d_g2.p, d_c.p, d_out.p);
  GF_HIP(hipGetLastError());
  std::vector<uint8_t> c8((size_t)n);
  GF_HIP(hipMemcpy(c8.data(), d_c.p, (size_t)n, hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(out_matches, d_out.p, 2 * (size_t)n * sizeof(gf_seqmatch), hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < n; ++r) out_counts[r] = c8[(size_t)r];
  return GF_OK;
}

// The exact wave-per-read kernels over a whole batch (gf_map_kernels.h): one launch per read-length class present
// (<= 256, <= 1024, <= 4096 bases), each skipping the reads of the other classes.  producer: 0 = probe every window,
// 1 = seed + verify (the <= 256 class).  top = the batch's highest class.
static void launch_wave_per_read(const GfTable& T_in, hipStream_t st, int n_cus, const uint8_t* bases, const int64_t* offsets, int64_t n,
                                 int top, int producer, uint8_t* counts, gf_seqmatch* matches) {
  // a zero-copy call's completion word (GfTable::done_*) is stored by the LAST launch only: the stream runs them in order
  GfTable Tq = T_in;
  Tq.done_ctr = nullptr;  // (done_flag stays: the earlier launches write a zero-copy call's results the same way)
  {
    const GfTable& T = top == 0 ? T_in : Tq;
    constexpr int W = 4;
    int grid = (int)std::min<int64_t>((n + W - 1) / W, (int64_t)n_cus * 8);
    if (producer == 0)
      hipLaunchKernelGGL((gf_k_map_reads_short<W, 0>), dim3(grid), dim3(W * 64), 0, st, T, bases, offsets,
                         n, top == 0 ? 1 : 0, counts, matches);
    else
      hipLaunchKernelGGL((gf_k_map_reads_short<W, 1>), dim3(grid), dim3(W * 64), 0, st, T, bases, offsets,
                         n, top == 0 ? 1 : 0, counts, matches);
  }
  if (top >= 1) {
    const GfTable& T = top == 1 ? T_in : Tq;
    constexpr int W = 4;
    int grid = (int)std::min<int64_t>((n + W - 1) / W, (int64_t)n_cus * 4 * 2);
    hipLaunchKernelGGL((gf_k_map_reads<1024, W, 0>), dim3(grid), dim3(W * 64), 0, st, T, bases, offsets,
                       n, 256, top == 1 ? 1 : 0, counts, matches);
  }
  if (top >= 2) {
    constexpr int W = 2;
    int grid = (int)std::min<int64_t>((n + W - 1) / W, (int64_t)n_cus * 2 * 2);
    hipLaunchKernelGGL((gf_k_map_reads<4096, W, 0>), dim3(grid), dim3(W * 64), 0, st, T_in, bases, offsets,
                       n, 1024, 1, counts, matches);
  }
}

static int map_span_device(const gf_index* idx, const ReadSrc& src, const int64_t* offsets, int64_t n,
                           int32_t max_read_len, uint8_t* counts, gf_seqmatch* matches, hipStream_t st, bool prof,
                           const int32_t* skip, int32_t fixed_len = 0, const int64_t* n_dev = nullptr) {
  gf_index* mix = const_cast<gf_index*>(idx);
  const uint8_t* bases = src.bases;
  if (src.packed() && idx->map_variant != 0)
    return fail(GF_ERR_ARG, "packed reads are taken by the flat pipeline only (gf_set_map_variant 0)");
  GfTable T = idx->table;
  T.skip = skip;  // (per call: the index itself stays read-only)
  T.fixed_len = fixed_len;
  T.n_dev = n_dev;
  if (fixed_len > 0 && (idx->map_variant != 0 || fixed_len > 320))
    return fail(GF_ERR_ARG, "fixed-length batches are taken by the flat pipeline only (reads of up to 320 bases, gf_set_map_variant 0)");
  // persistent grid: enough waves to fill every CU, reads interleaved across waves
  // One launch per read-length class present in the batch (<=256, <=1024, <=4096);
  // each launch skips the reads of the other classes, so short reads always get the
  // small-LDS kernel with the seed+verify first pass.
  const int top = max_read_len <= 256 ? 0 : (max_read_len <= 1024 ? 1 : 2);
  if (idx->map_variant == 0) {
    // flat pipeline (packing fused into seed+verify through LDS) for reads of up to 320 bases —
    // that covers the merged reads of 2 x 150 — and lists for the wave-per-read kernels of longer reads
    const int lmax = std::min<int>(max_read_len, 320);
    const int pw = lmax <= 160 ? 10 : (lmax <= 256 ? 16 : 20);
    const std::shared_ptr<Workspace> wse = ws_pool(false).entry(idx->device, st);
    std::lock_guard<std::mutex> ws_lock(wse->mu);  // held until this call's launches are queued
    const FlatPlan p = flat_plan(n, idx->n_cus, pw, max_read_len > lmax);
    void* ws_base = nullptr;
    int wrc = acquire_workspace(*wse, st, p.bytes(), &ws_base);
    if (wrc != GF_OK) return wrc;
    const FlatWs w = flat_carve((uint8_t*)ws_base, p);
    hipEvent_t* ev = prof ? mix->ev_stage : nullptr;
    if (src.packed())
      wrc = pw == 10   ? launch_flat<10, true>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev)
            : pw == 16 ? launch_flat<16, true>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev)
                       : launch_flat<20, true>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev);
    else
      wrc = pw == 10   ? launch_flat<10, false>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev)
            : pw == 16 ? launch_flat<16, false>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev)
                       : launch_flat<20, false>(idx, T, st, src, offsets, n, lmax, max_read_len, counts, matches, w, p, ev);
    if (wrc != GF_OK) return wrc;
    if (prof) mix->stages_recorded = true;
  } else {
    launch_wave_per_read(T, st, idx->n_cus, bases, offsets, n, top, idx->map_variant == 1 ? 0 : 1, counts, matches);
  }
  GF_HIP(hipGetLastError());
  return GF_OK;
}

// Reads per span of one gf_map_reads_device call.  The flat pipeline's lists hold read indices
// as uint32 and its workspace grows with the span (64-112 bytes per read), so a batch larger
// than this is mapped span by span on the same stream — same results, bounded workspace.
// GF_SPAN_MAX (environment) lowers it for the tests of the split itself.
static int64_t span_max_reads() {
  static const int64_t v = [] {
    int64_t d = (int64_t)1 << 30;
    if (const char* e = getenv("GF_SPAN_MAX")) {
      const long long x = atoll(e);
      if (x >= 1 && x < d) d = (int64_t)x;
    }
    return d;
  }();
  return v;
}

static int map_reads_device_impl(const gf_index* idx, const void* d_bases, const void* d_offsets, int64_t n,
                                 int32_t max_read_len, void* d_counts, void* d_matches, void* stream,
                                 const int32_t* d_skip, const void* d_pk = nullptr, const void* d_iv = nullptr,
                                 int32_t fixed_len = 0, const int64_t* d_n_actual = nullptr) {
  if (!idx || n < 0) return fail(GF_ERR_ARG, "null index or negative n");
  if (n == 0) return GF_OK;
  if ((!d_offsets && fixed_len <= 0) || !d_counts || !d_matches) return fail(GF_ERR_ARG, "null device pointer");
  if (max_read_len > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "max_read_len exceeds GF_MAX_READ_LEN");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  gf_index* mix = const_cast<gf_index*>(idx);
  const bool prof = idx->profiling;
  if (prof) {
    std::lock_guard<std::mutex> lk(mix->prof_mu);
    if (!mix->have_events) {
      GF_HIP(hipEventCreate(&mix->ev0));
      GF_HIP(hipEventCreate(&mix->ev1));
      for (auto& e : mix->ev_stage) GF_HIP(hipEventCreate(&e));
      mix->have_events = true;
    }
    GF_HIP(hipEventRecord(mix->ev0, st));
  }
  ReadSrc src;
  src.bases = (const uint8_t*)d_bases;
  src.pk = (const uint32_t*)d_pk;
  src.iv = (const uint16_t*)d_iv;
  const int64_t* offsets = (const int64_t*)d_offsets;
  uint8_t* counts = (uint8_t*)d_counts;
  gf_seqmatch* matches = (gf_seqmatch*)d_matches;
  const int64_t span = span_max_reads();
  for (int64_t s0 = 0; s0 < n; s0 += span) {
    const int64_t ns = std::min(span, n - s0);
    // offsets are absolute positions in `bases`: a span is the same call on a later part of the arrays
    ReadSrc span_src = src;
    if (fixed_len > 0) {  // a span of a fixed-length batch is the same call on later bases: read indices restart at 0
      if (span_src.bases) span_src.bases += s0 * (int64_t)fixed_len;
      if (span_src.packed() && ((s0 * (int64_t)fixed_len) & 15)) return fail(GF_ERR_ARG, "packed fixed-length batch beyond one span");
    }
    const int rc = map_span_device(idx, span_src, fixed_len > 0 ? nullptr : offsets + s0, ns, max_read_len, counts + s0,
                                   matches + 2 * s0, st, prof, d_skip ? d_skip + s0 : nullptr, fixed_len,
                                   n <= span ? d_n_actual : nullptr);  // (the count is of the whole batch: one span)
    if (rc != GF_OK) return rc;
  }
  if (prof) {
    std::lock_guard<std::mutex> lk(mix->prof_mu);
    GF_HIP(hipEventRecord(mix->ev1, st));
    mix->recorded = true;
  }
  return GF_OK;
}

int gf_segment_mask_test(const gf_index* idx, const uint8_t* masks, const int64_t* offsets, int64_t n,
                         const int64_t* gp1, const int64_t* gp2, int32_t* out_counts, gf_seqmatch* out_matches) {
  if (!idx || n < 0) return fail(GF_ERR_ARG, "null index or negative n");
  if (n == 0) return GF_OK;
  if (!offsets || !gp1 || !gp2 || !out_counts || !out_matches) return fail(GF_ERR_ARG, "null argument");
  for (int64_t r = 0; r < n; ++r) {
    const int64_t l = offsets[r + 1] - offsets[r];
    if (l < 1 || l > GF_MAX_READ_LEN) return fail(GF_ERR_ARG, "mask length must be 1..GF_MAX_READ_LEN");
  }
  if (offsets[0] != 0 || !masks) return fail(GF_ERR_ARG, "masks must start at offset 0");
  DeviceGuard guard(idx->device);
  return segment_mask_test_impl(idx, masks, offsets, n, gp1, gp2, out_counts, out_matches);
}

int64_t gf_packed_chunks(int64_t n_bases) { return n_bases < 0 ? 0 : (n_bases + 15) / 16 + 4; }

int gf_pack_bases_device(const gf_index* idx, const void* d_bases, int64_t n_bases, void* d_pk, void* d_iv, void* stream) {
  if (!idx || n_bases < 0) return fail(GF_ERR_ARG, "null index or negative size");
  if (!d_pk || !d_iv || (n_bases > 0 && !d_bases)) return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  const int64_t chunks = gf_packed_chunks(n_bases);  // (the padding chunks are written too: all "bad")
  const int grid = (int)std::min<int64_t>((chunks + 255) / 256, (int64_t)idx->n_cus * 32);
  hipLaunchKernelGGL(gf_k_pack_bases, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_bases, n_bases,
                     (uint32_t*)d_pk, (uint16_t*)d_iv, chunks);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_pack_bases_host(const char* bases, int64_t n_bases, uint32_t* pk, uint16_t* iv, int32_t n_threads) {
  if (n_bases < 0 || !pk || !iv || (n_bases > 0 && !bases)) return fail(GF_ERR_ARG, "bad argument");
  const int64_t chunks = gf_packed_chunks(n_bases);
  const unsigned char* b = (const unsigned char*)bases;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads <= 0 ? 1 : n_threads, chunks / 65536 + 1));
  if (T == 1) {
    gf_host_pack_range(b, n_bases, 0, chunks, pk, iv);
    return GF_OK;
  }
  std::vector<std::thread> th;
  const int64_t per = ((chunks + T - 1) / T + 1) & ~(int64_t)1;  // (even: the AVX2 loop takes chunks in pairs)
  for (int t = 0; t < T; ++t) {
    const int64_t c0 = std::min<int64_t>((int64_t)t * per, chunks), c1 = std::min<int64_t>(c0 + per, chunks);
    if (c0 < c1) th.emplace_back([=] { gf_host_pack_range(b, n_bases, c0, c1, pk, iv); });
  }
  for (auto& x : th) x.join();
  return GF_OK;
}

int gf_map_reads_packed_device(const gf_index* idx, const void* d_pk, const void* d_iv, const void* d_offsets, int64_t n,
                               int32_t max_read_len, void* d_counts, void* d_matches, void* stream) {
  if (n > 0 && (!d_pk || !d_iv)) return fail(GF_ERR_ARG, "null packed stream");
  return map_reads_device_impl(idx, nullptr, d_offsets, n, max_read_len, d_counts, d_matches, stream, nullptr, d_pk, d_iv);
}

int gf_map_reads_device(const gf_index* idx, const void* d_bases, const void* d_offsets, int64_t n,
                        int32_t max_read_len, void* d_counts, void* d_matches, void* stream) {
  return map_reads_device_impl(idx, d_bases, d_offsets, n, max_read_len, d_counts, d_matches, stream, nullptr);
}

int gf_map_reads_fixed_device(const gf_index* idx, const void* d_bases, int64_t n, int32_t read_len, void* d_counts,
                              void* d_matches, void* stream) {
  if (read_len < 1 || read_len > 320) return fail(GF_ERR_ARG, "read_len must be 1..320 for a fixed-length batch");
  if (n > 0 && !d_bases) return fail(GF_ERR_ARG, "null device pointer");
  return map_reads_device_impl(idx, d_bases, nullptr, n, read_len, d_counts, d_matches, stream, nullptr, nullptr, nullptr,
                               read_len);
}

int64_t gf_compact_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  int64_t ntiles = (n + GF_CTILE - 1) / GF_CTILE;
  return 64 + ntiles * (int64_t)(sizeof(uint32_t) + sizeof(int64_t)) + 16;
}

int gf_compact_hits_device(const gf_index* idx, const void* d_counts, const void* d_matches, int64_t n,
                           int64_t read_id_base, void* d_hits, int64_t hits_cap, void* d_n_hits,
                           void* d_workspace, void* stream) {
  if (!idx || n < 0 || hits_cap < 0) return fail(GF_ERR_ARG, "bad argument");
  if (!d_n_hits || !d_workspace) return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (n + GF_CTILE - 1) / GF_CTILE;
  // workspace: [int64 tile_offsets[ntiles]] [uint32 tile_counts[ntiles]]
  uintptr_t w = ((uintptr_t)d_workspace + 15) & ~(uintptr_t)15;
  int64_t* tile_offsets = (int64_t*)w;
  uint32_t* tile_counts = (uint32_t*)(w + (size_t)ntiles * sizeof(int64_t));
  if (ntiles > 0) {
    if (!d_counts || !d_matches || (hits_cap > 0 && !d_hits)) return fail(GF_ERR_ARG, "null device pointer");
    hipLaunchKernelGGL(gf_k_compact_count, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st,
                       (const uint8_t*)d_counts, n, tile_counts);
    GF_HIP(hipGetLastError());
  }
  launch_scan(st, ntiles, tile_counts, tile_offsets, (int64_t*)d_n_hits);
  GF_HIP(hipGetLastError());
  if (ntiles > 0) {
    hipLaunchKernelGGL(gf_k_compact_write, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st,
                       (const uint8_t*)d_counts, (const gf_seqmatch*)d_matches, n, read_id_base, tile_offsets,
                       (gf_hit*)d_hits, hits_cap);
    GF_HIP(hipGetLastError());
  }
  return GF_OK;
}

// ---- host-buffer entry points ---------------------------------------------------------------
// A host-buffer call (gf_map_reads, gf_map_read, gf_map_reads_hits) borrows a LANE of the index: a
// stream of its own, a grow-only device arena (bases, offsets, counts, matches, hits, compaction
// workspace) and a small pinned block for what comes back.  Lanes are what makes the boundary
// usable the way the reference uses Indexer::map_read — from t-1 consumer threads on one `&self`
// (pescanner.rs:296-311): concurrent callers get different lanes, so their copies and kernels
// overlap instead of queueing behind one mutex; a caller waits only when all lanes are taken.
// (For a PCIe-bound path the per-call hipMalloc/hipFree of gigabyte buffers cost as much as the
// copies: hence the arenas.)
// Blocks handed out by gf_host_alloc: a zero-copy call whose reads lie in one of them lets its kernel read them where
// they are (no staging memcpy).  Few entries, looked up under a shared lock.
struct PinnedBlock {
  const char* host;
  size_t bytes;
  const uint8_t* dev;
};
static std::vector<PinnedBlock> g_pinned;
static std::shared_mutex g_pinned_mu;
static const uint8_t* pinned_device_address(const char* p, size_t len) {
  std::shared_lock<std::shared_mutex> lk(g_pinned_mu);
  for (const PinnedBlock& b : g_pinned)
    if (p >= b.host && p + len <= b.host + b.bytes) return b.dev + (p - b.host);
  return nullptr;
}

// A zero-copy call: reads staged in (or read straight from) pinned host memory, ONE launch of the exact
// wave-per-read kernels that fetches them over the link and writes counts and matches back into the same pinned
// block, completion reported by a word the kernel's last block stores (GfTable::done_*) — no copy commands, no
// stream synchronisation.  A lane of the host-buffer calls and a slot of a gf_stream each own one.
struct ZeroCopy {
  uint8_t* pinned = nullptr;    // host: | completion word (64 B) | offsets | matches | counts | bases |
  uint8_t* d_pinned = nullptr;  // the same block as the device addresses it
  size_t pinned_bytes = 0;
  unsigned int* d_done_ctr = nullptr;  // device: the blocks count themselves out here
  uint32_t seq = 0;                    // number of the last call = the value its kernel stores to the completion word
  uint32_t unsynced = 0;               // calls since the stream was last synchronised
  // the call in flight (zc_submit .. zc_wait)
  const uint8_t* h_counts = nullptr;
  const gf_seqmatch* h_matches = nullptr;
  bool flag_wait = false;
  void release() {
    if (pinned) (void)hipHostFree(pinned);
    if (d_done_ctr) (void)hipFree(d_done_ctr);
    pinned = d_pinned = nullptr;
    d_done_ctr = nullptr;
    pinned_bytes = 0;
  }
};

struct HostLane {
  hipStream_t st = nullptr;
  void* arena = nullptr;
  size_t arena_bytes = 0;
  ZeroCopy zc;  // the reads and results of zero-copy calls
  bool busy = false;
};

struct LaneLease {
  gf_index* ix;
  HostLane* lane = nullptr;
  explicit LaneLease(gf_index* i) : ix(i) {}
  int acquire();
  ~LaneLease();
};

static const int GF_MAX_LANES = 8;

void gf_index::free_lanes() {  // (the destructor has waited for the device)
  for (HostLane* L : lanes) {
    if (L->arena) (void)hipFree(L->arena);
    L->zc.release();
    if (L->st) {
      // every host-buffer call went through gf_map_reads_device(L->st): its mapping workspace (64-112 B per read of
      // the largest batch) is keyed by this stream in the process-wide pool and would outlive it — a host that
      // rebuilds the index per CSV left one multi-GB workspace behind per freed index
      ws_pool(false).drop(device, L->st);
      ws_pool(true).drop(device, L->st);
      (void)hipStreamDestroy(L->st);
    }
    delete L;
  }
  lanes.clear();
}

int LaneLease::acquire() {
  std::unique_lock<std::mutex> lk(ix->lane_mu);
  for (;;) {
    for (HostLane* L : ix->lanes)
      if (!L->busy) {
        L->busy = true;
        lane = L;
        return GF_OK;
      }
    if ((int)ix->lanes.size() < GF_MAX_LANES) {
      HostLane* L = new HostLane();
      hipError_t e = hipStreamCreateWithFlags(&L->st, hipStreamNonBlocking);
      if (e != hipSuccess) {
        delete L;
        return fail(GF_ERR_HIP, std::string("hipStreamCreateWithFlags failed: ") + hipGetErrorString(e));
      }
      L->busy = true;
      ix->lanes.push_back(L);
      lane = L;
      return GF_OK;
    }
    ix->lane_cv.wait(lk);  // every lane is taken by another caller
  }
}

LaneLease::~LaneLease() {
  if (!lane) return;
  {
    std::lock_guard<std::mutex> lk(ix->lane_mu);
    lane->busy = false;
  }
  ix->lane_cv.notify_one();
}

struct HostStage {
  uint8_t* bases = nullptr;     // device pointer such that bases + offsets[r] is read r (host offsets kept as they are)
  int64_t* offsets = nullptr;
  uint8_t* counts = nullptr;
  gf_seqmatch* matches = nullptr;
  gf_hit* hits = nullptr;
  int64_t* total = nullptr;
  uint8_t* compact_ws = nullptr;
};

static int lane_reserve(HostLane& L, size_t need) {
  if (L.arena_bytes >= need) return GF_OK;
  if (L.arena) {
    GF_HIP(hipStreamSynchronize(L.st));
    GF_HIP(hipFree(L.arena));
    L.arena = nullptr;
    L.arena_bytes = 0;
  }
  const size_t want = need + need / 4;  // (slack: see acquire_workspace)
  if (hipMalloc(&L.arena, want) == hipSuccess) {
    L.arena_bytes = want;
  } else {
    (void)hipGetLastError();
    GF_HIP(hipMalloc(&L.arena, need));
    L.arena_bytes = need;
  }
  return GF_OK;
}

static int zc_reserve(ZeroCopy& Z, hipStream_t st, size_t need) {
  if (!Z.d_done_ctr) {
    GF_HIP(hipMalloc((void**)&Z.d_done_ctr, 256));
    GF_HIP(hipMemset(Z.d_done_ctr, 0, 256));
  }
  if (Z.pinned_bytes >= need) return GF_OK;
  if (Z.pinned) {
    GF_HIP(hipStreamSynchronize(st));
    GF_HIP(hipHostFree(Z.pinned));
    Z.pinned = nullptr;
    Z.pinned_bytes = 0;
  }
  need = std::max<size_t>(need + need / 4, 64 * 1024);
  GF_HIP(hipHostMalloc((void**)&Z.pinned, need, hipHostMallocDefault));
  Z.pinned_bytes = need;
  void* dptr = nullptr;
  GF_HIP(hipHostGetDevicePointer(&dptr, Z.pinned, 0));
  Z.d_pinned = (uint8_t*)dptr;
  *(volatile uint32_t*)Z.pinned = Z.seq;  // the completion word (first 64 bytes of the block)
  return GF_OK;
}

// validates the batch, sizes the lane's arena, queues the copies in and the mapping on the lane's stream
static int stage_and_map(gf_index* mix, HostLane& L, const char* bases, const int64_t* offsets, int64_t n,
                         int64_t hits_cap, HostStage& S) {
  if (n < 0) return fail(GF_ERR_ARG, "negative n");
  if (n > 0 && (!offsets)) return fail(GF_ERR_ARG, "offsets is null");
  int64_t maxlen = 0;
  for (int64_t r = 0; r < n; ++r) {
    int64_t l = offsets[r + 1] - offsets[r];
    if (l < 0) return fail(GF_ERR_ARG, "offsets must be non-decreasing");
    maxlen = std::max(maxlen, l);
  }
  if (maxlen > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "a read exceeds GF_MAX_READ_LEN");
  const int64_t b0 = n > 0 ? offsets[0] : 0, b1 = n > 0 ? offsets[n] : 0;
  if (b1 > b0 && !bases) return fail(GF_ERR_ARG, "bases is null");
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t sz_bases = al((size_t)(b1 - b0) + 64), sz_off = al(((size_t)n + 1) * sizeof(int64_t));
  const size_t sz_counts = al((size_t)n + 1), sz_matches = al(((size_t)n * 2 + 1) * sizeof(gf_seqmatch));
  const size_t sz_hits = al(((size_t)hits_cap + 1) * sizeof(gf_hit)), sz_total = 256;
  const size_t sz_cws = al((size_t)gf_compact_workspace_bytes(n) + 16);
  int rc = lane_reserve(L, sz_bases + sz_off + sz_counts + sz_matches + sz_hits + sz_total + sz_cws);
  if (rc != GF_OK) return rc;
  uint8_t* wp = (uint8_t*)L.arena;
  uint8_t* d_bases = wp; wp += sz_bases;
  S.offsets = (int64_t*)wp; wp += sz_off;
  S.counts = wp; wp += sz_counts;
  S.matches = (gf_seqmatch*)wp; wp += sz_matches;
  S.hits = (gf_hit*)wp; wp += sz_hits;
  S.total = (int64_t*)wp; wp += sz_total;
  S.compact_ws = wp;
  // the 16-byte boundary at or below the first read is preserved (the kernels stage whole chunks)
  const size_t lead = (size_t)((uintptr_t)(bases + b0) & 15u);
  S.bases = d_bases + lead - b0;  // S.bases + offsets[r] = the device copy of read r
  if (n == 0) return GF_OK;
  if (b1 > b0) GF_HIP(hipMemcpyAsync(d_bases + lead, bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, L.st));
  GF_HIP(hipMemcpyAsync(S.offsets, offsets, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, L.st));
  return gf_map_reads_device(mix, S.bases, S.offsets, n, (int32_t)std::max<int64_t>(maxlen, 1), S.counts, S.matches,
                             (void*)L.st);
}

// Calls of a few reads — Indexer::map_read as the reference calls it, one read at a time — and calls of one PACK
// of the reference's size (PACK_SIZE = 1000 pairs, common.rs:23; pescanner.rs:427-518) skip the copies altogether:
// the reads are written to the lane's pinned block, the exact wave-per-read kernels fetch them over the link and
// write their results back to the same block: ONE launch and one wait per call where the batch route is two
// copies in, a memset, seven launches and two copies out — 12 queue operations whose cost, not the kernels', is
// what a 2000-read call pays for (tools/bench_pack_sweep.py; DESIGN.md 4, "host calls").
//   n <= GF_SMALL_CALL_READS (64):     one wave per read, probe-all kernel, a block per read
//   n <= GF_PACK_CALL_READS:           the software-pipelined seed+verify kernel, grid-stride over the pack
static int64_t small_call_reads() {
  static const int64_t v = [] {
    int64_t d = 64;
    if (const char* e = getenv("GF_SMALL_CALL_READS")) d = std::max<int64_t>(0, atoll(e));  // experiments
    return d;
  }();
  return v;
}
#define GF_SMALL_CALL_READS small_call_reads()
#ifndef GF_PACK_CALL_READS_DEFAULT
#define GF_PACK_CALL_READS_DEFAULT 8192  // (reads: packs of up to 4096 pairs; the crossover is measured by tools/bench_pack_sweep.py)
#endif
static int64_t pack_call_reads() {
  static const int64_t v = [] {
    int64_t d = GF_PACK_CALL_READS_DEFAULT;
    if (const char* e = getenv("GF_PACK_CALL_READS")) d = std::max<int64_t>(0, atoll(e));  // 0 = the batch route for every call beyond the small ones
    return d;
  }();
  return v;
}
static inline bool zero_copy_call(const gf_index* idx, int64_t n) {
  const int64_t lim = idx->pack_call_reads >= 0 ? idx->pack_call_reads : pack_call_reads();
  return idx->map_variant == 0 && n > 0 && (n <= GF_SMALL_CALL_READS || n <= lim);
}

// validates the pack, stages it, queues its one launch on `st`
static int zc_submit(gf_index* mix, ZeroCopy& Z, hipStream_t st, const char* bases, const int64_t* offsets, int64_t n) {
  int64_t maxlen = 0;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t l = offsets[r + 1] - offsets[r];
    if (l < 0) return fail(GF_ERR_ARG, "offsets must be non-decreasing");
    maxlen = std::max(maxlen, l);
  }
  if (maxlen > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "a read exceeds GF_MAX_READ_LEN");
  const int64_t b0 = offsets[0], b1 = offsets[n];
  if (b1 > b0 && !bases) return fail(GF_ERR_ARG, "bases is null");
  // reads in a gf_host_alloc block are read where they are; anything else is staged in the pinned block
  const uint8_t* d_src = b1 > b0 ? pinned_device_address(bases + b0, (size_t)(b1 - b0)) : nullptr;
  const size_t sz_flag = 64, sz_off = (((size_t)n + 1) * 8 + 15) & ~(size_t)15, sz_cnt = ((size_t)n + 15) & ~(size_t)15, sz_m = (size_t)n * 32;
  const size_t sz_b = d_src ? 0 : (((size_t)(b1 - b0) + 64 + 15) & ~(size_t)15);
  int rc = zc_reserve(Z, st, sz_flag + sz_off + sz_cnt + sz_m + sz_b + 64);
  if (rc != GF_OK) return rc;
  int64_t* h_off = (int64_t*)(Z.pinned + sz_flag);
  gf_seqmatch* h_m = (gf_seqmatch*)(Z.pinned + sz_flag + sz_off);
  uint8_t* h_cnt = Z.pinned + sz_flag + sz_off + sz_m;
  uint8_t* h_b = Z.pinned + sz_flag + sz_off + sz_m + sz_cnt;
  for (int64_t r = 0; r <= n; ++r) h_off[r] = offsets[r] - b0;
  if (!d_src && b1 > b0) memcpy(h_b, bases + b0, (size_t)(b1 - b0));
  memset(h_cnt, 0, (size_t)n);  // (the kernels write the counts of the reads with segments only: GfTable::done_flag)
  uint8_t* d = Z.d_pinned;
  GfTable T = mix->table;
  static const bool flag_wait = !(getenv("GF_PACK_WAIT") && !strcmp(getenv("GF_PACK_WAIT"), "sync"));  // experiments
  Z.flag_wait = flag_wait;
  Z.seq += 1;
  if (flag_wait) {
    T.done_ctr = Z.d_done_ctr;
    T.done_flag = (unsigned int*)d;
    T.done_seq = Z.seq;
  }
  const uint8_t* d_b = d_src ? d_src : (const uint8_t*)(d + (h_b - Z.pinned));
  const int64_t* d_off = (const int64_t*)(d + sz_flag);
  uint8_t* d_cnt = d + (h_cnt - Z.pinned);
  gf_seqmatch* d_m = (gf_seqmatch*)(d + sz_flag + sz_off);
  if (n <= GF_SMALL_CALL_READS) {
    const int grid = (int)n;
    if (maxlen <= 256)
      hipLaunchKernelGGL((gf_k_map_reads<256, 1, 0>), dim3(grid), dim3(64), 0, st, T, d_b, d_off, n, -1, 1, d_cnt, d_m);
    else if (maxlen <= 1024)
      hipLaunchKernelGGL((gf_k_map_reads<1024, 1, 0>), dim3(grid), dim3(64), 0, st, T, d_b, d_off, n, -1, 1, d_cnt, d_m);
    else
      hipLaunchKernelGGL((gf_k_map_reads<4096, 1, 0>), dim3(grid), dim3(64), 0, st, T, d_b, d_off, n, -1, 1, d_cnt, d_m);
  } else {
    const int top = maxlen <= 256 ? 0 : (maxlen <= 1024 ? 1 : 2);
    launch_wave_per_read(T, st, mix->n_cus, d_b, d_off, n, top, 1, d_cnt, d_m);
  }
  GF_HIP(hipGetLastError());
  Z.h_counts = h_cnt;
  Z.h_matches = h_m;
  return GF_OK;
}

// waits for the call queued by zc_submit: on the word the kernel's last block stores — no queue operation, no
// runtime lock shared with the other callers — and, now and then, by a stream synchronisation that lets the
// runtime retire the launches; a kernel that never reports (a fault) is found by that synchronisation too
static int zc_wait(ZeroCopy& Z, hipStream_t st) {
  bool done = false;
  const uint32_t* h_flag = (const uint32_t*)Z.pinned;
  if (Z.flag_wait) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 1;; ++spin) {
      if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == Z.seq) { done = true; break; }
      __builtin_ia32_pause();
      // a kernel that takes longer than a pack's few tens of microseconds (the device is busy with somebody's batch):
      // the core goes to whoever else wants it between looks
      if (spin > 4096) std::this_thread::yield();
      if ((spin & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
    if (done && ++Z.unsynced >= 64) done = false;
  }
  if (!done) {
    GF_HIP(hipStreamSynchronize(st));
    Z.unsynced = 0;
    if (Z.flag_wait && __atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != Z.seq)
      return fail(GF_ERR_HIP, "a zero-copy call's kernel ended without reporting completion");
  }
  return GF_OK;
}

// the hit records of a finished zero-copy call, put together on the host as gf_k_compact_write does on the device
static int64_t zc_hits(const ZeroCopy& Z, int64_t n, int64_t read_id_base, gf_hit* out_hits, int64_t cap) {
  int64_t total = 0;
  const uint8_t* c8 = Z.h_counts;
  const gf_seqmatch* m = Z.h_matches;
  for (int64_t r = 0; r < n; ++r) {
    const int c = c8[r];
    if (c != 1 && c != 2) continue;
    if (total < cap) {
      gf_hit& h = out_hits[total];
      h.read_id = read_id_base + r;
      h.n = c;
      h.pad = 0;
      h.m[0] = m[2 * r];
      if (c > 1) h.m[1] = m[2 * r + 1]; else memset(&h.m[1], 0, sizeof(gf_seqmatch));
    }
    total += 1;
  }
  return total;
}

static int map_small(gf_index* mix, HostLane& L, const char* bases, const int64_t* offsets, int64_t n,
                     const uint8_t** out_counts, const gf_seqmatch** out_matches) {
  int rc = zc_submit(mix, L.zc, L.st, bases, offsets, n);
  if (rc != GF_OK) return rc;
  rc = zc_wait(L.zc, L.st);
  if (rc != GF_OK) return rc;
  *out_counts = L.zc.h_counts;
  *out_matches = L.zc.h_matches;
  return GF_OK;
}

int gf_map_reads(const gf_index* idx, const char* bases, const int64_t* offsets, int64_t n,
                 int32_t* out_counts, gf_seqmatch* out_matches) {
  if (n > 0 && (!out_counts || !out_matches)) return fail(GF_ERR_ARG, "null output buffer");
  if (!idx) return fail(GF_ERR_ARG, "null index");
  if (n < 0) return fail(GF_ERR_ARG, "negative n");
  if (n > 0 && !offsets) return fail(GF_ERR_ARG, "offsets is null");
  if (n == 0) return GF_OK;
  DeviceGuard guard(idx->device);
  gf_index* mix = const_cast<gf_index*>(idx);
  LaneLease lease(mix);
  int rc = lease.acquire();
  if (rc != GF_OK) return rc;
  HostLane& L = *lease.lane;
  if (zero_copy_call(idx, n)) {
    const uint8_t* c8 = nullptr;
    const gf_seqmatch* m = nullptr;
    rc = map_small(mix, L, bases, offsets, n, &c8, &m);
    if (rc != GF_OK) return rc;
    for (int64_t r = 0; r < n; ++r) {
      const int c = c8[r];
      out_counts[r] = c;
      for (int k = 0; k < c && k < 2; ++k) out_matches[2 * r + k] = m[2 * r + k];
    }
    return GF_OK;
  }
  // the dense counts come back whole (a byte per read); of the matches only the reads that have
  // any (ordered compaction on the device): 48 bytes per hit instead of 32 per read
  const int64_t cap = std::max<int64_t>(1024, n / 16);
  HostStage S;
  // Whatever way this call ends, nothing may still be in flight on the lane when it does: the copies below land in
  // local pageable buffers, and the lane goes back to the pool (an error between an asynchronous copy
  // and its synchronisation left both dangling).
  struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{L.st};
  rc = stage_and_map(mix, L, bases, offsets, n, cap, S);
  if (rc != GF_OK) return rc;
  rc = gf_compact_hits_device(idx, S.counts, S.matches, n, 0, S.hits, cap, S.total, S.compact_ws, (void*)L.st);
  if (rc != GF_OK) return rc;
  std::vector<uint8_t> c8((size_t)n);
  int64_t total = 0;
  GF_HIP(hipMemcpyAsync(c8.data(), S.counts, (size_t)n, hipMemcpyDeviceToHost, L.st));
  GF_HIP(hipMemcpyAsync(&total, S.total, sizeof total, hipMemcpyDeviceToHost, L.st));
  GF_HIP(hipStreamSynchronize(L.st));
  for (int64_t r = 0; r < n; ++r) out_counts[r] = c8[(size_t)r];
  if (total <= cap) {
    std::vector<gf_hit> h((size_t)total);
    if (total > 0) {
      GF_HIP(hipMemcpyAsync(h.data(), S.hits, (size_t)total * sizeof(gf_hit), hipMemcpyDeviceToHost, L.st));
      GF_HIP(hipStreamSynchronize(L.st));
    }
    for (const gf_hit& x : h)
      for (int k = 0; k < x.n && k < 2; ++k) out_matches[2 * x.read_id + k] = x.m[k];
  } else {  // a batch of mostly hits: the dense array after all
    std::vector<gf_seqmatch> m((size_t)n * 2);
    GF_HIP(hipMemcpyAsync(m.data(), S.matches, (size_t)n * 2 * sizeof(gf_seqmatch), hipMemcpyDeviceToHost, L.st));
    GF_HIP(hipStreamSynchronize(L.st));
    for (int64_t r = 0; r < n; ++r)
      for (int k = 0; k < c8[(size_t)r] && k < 2; ++k) out_matches[2 * r + k] = m[(size_t)(2 * r + k)];
  }
  return GF_OK;
}

int gf_map_read(const gf_index* idx, const char* seq, int64_t len, gf_seqmatch out[2]) {
  if (!out || len < 0) return fail(GF_ERR_ARG, "bad argument");
  int64_t offsets[2] = {0, len};
  int32_t count = 0;
  gf_seqmatch m[2];
  int rc = gf_map_reads(idx, seq, offsets, 1, &count, m);
  if (rc != GF_OK) return rc;
  for (int k = 0; k < count; ++k) out[k] = m[k];
  return count;
}

int gf_map_reads_hits(const gf_index* idx, const char* bases, const int64_t* offsets, int64_t n,
                      int64_t read_id_base, gf_hit* out_hits, int64_t cap, int64_t* out_n) {
  if (!out_n || cap < 0 || (cap > 0 && !out_hits)) return fail(GF_ERR_ARG, "bad output argument");
  if (!idx) return fail(GF_ERR_ARG, "null index");
  *out_n = 0;
  DeviceGuard guard(idx->device);
  gf_index* mix = const_cast<gf_index*>(idx);
  LaneLease lease(mix);
  int rc = lease.acquire();
  if (rc != GF_OK) return rc;
  HostLane& L = *lease.lane;
  if (n < 0) return fail(GF_ERR_ARG, "negative n");
  if (n > 0 && !offsets) return fail(GF_ERR_ARG, "offsets is null");
  if (zero_copy_call(idx, n)) {  // one pack of the reference's size: one launch, the hit records put together here
    const uint8_t* c8 = nullptr;
    const gf_seqmatch* m = nullptr;
    rc = map_small(mix, L, bases, offsets, n, &c8, &m);
    if (rc != GF_OK) return rc;
    *out_n = zc_hits(L.zc, n, read_id_base, out_hits, cap);
    return GF_OK;
  }
  HostStage S;
  struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{L.st};  // (see gf_map_reads)
  rc = stage_and_map(mix, L, bases, offsets, n, cap, S);
  if (rc != GF_OK || n == 0) return rc;
  rc = gf_compact_hits_device(idx, S.counts, S.matches, n, read_id_base, S.hits, cap, S.total, S.compact_ws, (void*)L.st);
  if (rc != GF_OK) return rc;
  int64_t total = 0;
  GF_HIP(hipMemcpyAsync(&total, S.total, sizeof total, hipMemcpyDeviceToHost, L.st));
  GF_HIP(hipStreamSynchronize(L.st));  // (the launches are done)
  *out_n = total;
  int64_t ncopy = std::min(total, cap);
  if (ncopy > 0) {
    GF_HIP(hipMemcpyAsync(out_hits, S.hits, (size_t)ncopy * sizeof(gf_hit), hipMemcpyDeviceToHost, L.st));
    GF_HIP(hipStreamSynchronize(L.st));
  }
  return GF_OK;
}

// src/core/indexer.rs:541-608
int gf_in_required_direction(const gf_seqmatch* m, int32_t n, const uint8_t* gene_reversed, int32_t n_genes) {
  if (n < 2) return 0;
  if (!m || !gene_reversed) return fail(GF_ERR_ARG, "null argument");
  const gf_seqmatch* left = &m[0];
  const gf_seqmatch* right = &m[1];
  if (left->seq_start > right->seq_start) std::swap(left, right);
  if (left->position > 0 && right->position > 0) return 1;
  if (left->position < 0 && right->position < 0) return 0;
  if (left->contig < 0 || left->contig >= n_genes || right->contig < 0 || right->contig >= n_genes)
    return fail(GF_ERR_ARG, "contig out of range");
  const bool lrev = gene_reversed[left->contig] != 0, rrev = gene_reversed[right->contig] != 0;
  if (lrev && !rrev) return 0;
  if (!lrev && rrev) return 1;
  if (left->contig < right->contig) return 1;
  // the reference's same-contig test compares left with itself (:598) and is never true
  return 0;
}

// ---- FusionMapper::map_read tail, host logic --------------------

namespace {

// Hyyro's block-based bit-vector Levenshtein (the algorithm of edit_distance.rs:12-92):
// `a` is the pattern (blocks of 64 symbols), `b` the text.
size_t ed_bitparallel(const unsigned char* a, size_t asize, const unsigned char* b, size_t bsize) {
  const size_t tmax = (asize - 1) >> 6;      // index of the last block
  const size_t tlen = asize - tmax * 64;     // symbols in the last block
  const size_t nb = tmax + 1;
  std::vector<uint64_t> peq(256 * nb, 0);    // match masks per symbol and block
  for (size_t i = 0; i < asize; ++i) peq[(size_t)a[i] * nb + (i >> 6)] |= 1ull << (i & 63);
  std::vector<uint64_t> vp(nb, ~0ull), vn(nb, 0), hp(nb, 0), hn(nb, 0);
  vp[tmax] = tlen == 64 ? ~0ull : ((1ull << tlen) - 1);
  const uint64_t top = 1ull << (tlen - 1), msb = 1ull << 63;
  size_t d = asize;
  for (size_t i = 0; i < bsize; ++i) {
    const uint64_t* pm = &peq[(size_t)b[i] * nb];
    for (size_t r = 0; r < nb; ++r) {
      uint64_t x = pm[r];
      const bool carry_n = r > 0 && (hn[r - 1] & msb);
      if (carry_n) x |= 1ull;
      const uint64_t d0 = (((x & vp[r]) + vp[r]) ^ vp[r]) | x | vn[r];
      hp[r] = vn[r] | ~(d0 | vp[r]);
      hn[r] = d0 & vp[r];
      uint64_t y = hp[r] << 1;
      if (r == 0 || (hp[r - 1] & msb)) y |= 1ull;
      vp[r] = (hn[r] << 1) | ~(d0 | y);
      if (carry_n) vp[r] |= 1ull;
      vn[r] = d0 & y;
    }
    if (hp[tmax] & top) d += 1;
    else if (hn[tmax] & top) d -= 1;
  }
  return d;
}

size_t ed_dp(const unsigned char* a, size_t n1, const unsigned char* b, size_t n2) {
  std::vector<uint32_t> prev(n2 + 1), cur(n2 + 1);
  for (size_t j = 0; j <= n2; ++j) prev[j] = (uint32_t)j;
  for (size_t i = 1; i <= n1; ++i) {
    cur[0] = (uint32_t)i;
    for (size_t j = 1; j <= n2; ++j)
      cur[j] = std::min(std::min(prev[j], cur[j - 1]) + 1, prev[j - 1] + (a[i - 1] == b[j - 1] ? 0u : 1u));
    std::swap(prev, cur);
  }
  return prev[n2];
}

// edit_distance.rs:159-193: the longer string is the pattern unless it needs more than 10 blocks
size_t host_edit_distance(const unsigned char* a, size_t asize, const unsigned char* b, size_t bsize) {
  if (asize == 0) return bsize;
  if (bsize == 0) return asize;
  if (asize < bsize) { std::swap(a, b); std::swap(asize, bsize); }
  if (((asize - 1) >> 6) + 1 > 10) { std::swap(a, b); std::swap(asize, bsize); }
  if (((asize - 1) >> 6) + 1 <= 10) return ed_bitparallel(a, asize, b, bsize);
  return ed_dp(a, asize, b, bsize);
}

char host_complement(char c) {  // sequence.rs:51-59
  switch (c) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

// fusion_mapper.rs:225-251
int32_t host_calc_ed(const char* fs, int64_t fs_len, const char* seq, int32_t seq_len, int32_t start, int32_t end) {
  if ((start >= 0 && end <= 0) || (start <= 0 && end >= 0)) return -1;  // not on one strand
  if (std::abs((int64_t)start) >= fs_len || std::abs((int64_t)end) >= fs_len) return -2;
  std::string ss(seq, (size_t)seq_len);
  if (start < 0) {
    std::string rc((size_t)seq_len, 'N');
    for (int32_t i = 0; i < seq_len; ++i) rc[(size_t)(seq_len - 1 - i)] = host_complement(seq[i]);
    ss.swap(rc);
    const int32_t tmp = start;
    start = -end;
    end = -tmp;
  }
  return (int32_t)host_edit_distance((const unsigned char*)ss.data(), ss.size(), (const unsigned char*)fs + start,
                                     (size_t)(end - start + 1));
}

}  // namespace

int64_t gf_edit_distance(const char* a, int64_t alen, const char* b, int64_t blen) {
  if (alen < 0 || blen < 0 || (alen > 0 && !a) || (blen > 0 && !b)) return fail(GF_ERR_ARG, "bad argument");
  return (int64_t)host_edit_distance((const unsigned char*)a, (size_t)alen, (const unsigned char*)b, (size_t)blen);
}

// make_match (fusion_mapper.rs:154-194) + calc_distance (:196-251) for a two-segment mapping
static int gf_make_match_impl(const char* const* fusion_seqs, const int64_t* fusion_lens, int32_t n_genes, const char* seq,
                              int64_t len, const gf_seqmatch* mapping, gf_readmatch* out) {
  gf_seqmatch left = mapping[0], right = mapping[1];
  if (left.seq_start > right.seq_start) std::swap(left, right);
  if (left.contig < 0 || left.contig >= n_genes || right.contig < 0 || right.contig >= n_genes)
    return fail(GF_ERR_ARG, "contig out of range");
  const int32_t read_break = (left.seq_end + right.seq_start) / 2;  // :173
  left.position += read_break;                                       // :177-178
  right.position += read_break + 1;
  const int32_t left_len = read_break + 1, right_len = (int32_t)len - (read_break + 1);  // :199-201
  if (left_len < 0 || right_len < 0 || left_len > len) return fail(GF_ERR_ARG, "segments outside the read");
  out->read_break = read_break;
  out->gap = right.seq_start - left.seq_end - 1;                     // :180
  out->left_contig = left.contig;
  out->left_position = left.position;
  out->right_contig = right.contig;
  out->right_position = right.position;
  out->left_distance = host_calc_ed(fusion_seqs[left.contig], fusion_lens[left.contig], seq, left_len,
                                    left.position - left_len + 1, left.position);
  out->right_distance = host_calc_ed(fusion_seqs[right.contig], fusion_lens[right.contig], seq + read_break + 1,
                                     right_len, right.position, right.position + right_len - 1);
  return GF_RM_MATCH;
}

int gf_fusion_map_read(const char* const* fusion_seqs, const int64_t* fusion_lens, int32_t n_genes,
                       const uint8_t* gene_reversed, const char* seq, int64_t len, const gf_seqmatch* mapping,
                       int32_t n_mapping, gf_readmatch* out) {
  if (n_mapping < 0 || len < 0 || (n_mapping > 0 && !mapping)) return fail(GF_ERR_ARG, "bad argument");
  if (n_mapping < 2) return GF_RM_NONE;  // fusion_mapper.rs:107-115
  const int dir = gf_in_required_direction(mapping, n_mapping, gene_reversed, n_genes);
  if (dir < 0) return dir;
  if (!dir) return GF_RM_NONE_MAPABLE;   // :118-123
  if (n_mapping != 2) return GF_RM_NONE_MAPABLE;  // make_match returns None (:155-157)
  if (!out || !seq || !fusion_seqs || !fusion_lens) return fail(GF_ERR_ARG, "null argument");
  return gf_make_match_impl(fusion_seqs, fusion_lens, n_genes, seq, len, mapping, out);
}

// The host-side tail for a whole list of pair hits: FusionMapper::make_match + calc_distance
// (fusion_mapper.rs:154-251) per record, on n_threads host threads (records are independent).
int gf_pair_hits_finish(const gf_index* idx, const gf_pair_hit* hits, int64_t n, const char* hit_bases,
                        int64_t hit_bytes, gf_readmatch* out, int32_t* out_status, int32_t n_threads) {
  if (!idx || n < 0 || hit_bytes < 0) return fail(GF_ERR_ARG, "null index or negative size");
  if (n == 0) return GF_OK;
  if (!hits || !out || !out_status || (hit_bytes > 0 && !hit_bases)) return fail(GF_ERR_ARG, "null argument");
  const size_t ng = idx->fusion_seq.size();
  std::vector<const char*> fp(ng, nullptr);
  std::vector<int64_t> fl(ng);
  for (size_t c = 0; c < ng; ++c) fl[c] = (int64_t)idx->gene_len_h[c];
  for (int64_t k = 0; k < n; ++k)   // the genes the records name (the only ones make_match looks at): fetched once
    for (int e = 0; e < 2; ++e) {
      const int32_t c = hits[k].m[e].contig;
      if (c < 0 || (size_t)c >= ng) return fail(GF_ERR_ARG, "contig out of range");
      if (!fp[(size_t)c]) {
        if (int rc = idx->ensure_fusion(c)) return rc;
        fp[(size_t)c] = idx->fusion_seq[(size_t)c].data();
      }
    }
  for (int64_t k = 0; k < n; ++k)
    if (hits[k].seq_offset < 0 || hits[k].read_len < 0 || hits[k].seq_offset + hits[k].read_len > hit_bytes)
      return fail(GF_ERR_ARG, "a record's read lies outside hit_bases");
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads <= 0 ? 1 : n_threads, (n + 255) / 256));
  std::atomic<int> err{GF_OK};
  std::string err_msg;
  std::mutex err_mu;
  auto work = [&](int t) {
    for (int64_t k = t; k < n; k += T) {
      const gf_pair_hit& h = hits[k];
      // make_match (fusion_mapper.rs:154-194) and calc_distance (:196-251); the gate of :118-123 has
      // already passed on the device, so it is not asked again
      gf_seqmatch m[2] = {h.m[0], h.m[1]};
      int rc = gf_make_match_impl(fp.data(), fl.data(), (int32_t)ng, hit_bases + h.seq_offset, h.read_len, m, &out[k]);
      out_status[k] = rc;
      if (rc < 0) {
        std::lock_guard<std::mutex> lk(err_mu);
        err = rc;
        err_msg = g_err;
        return;
      }
    }
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  if (err.load() != GF_OK) return fail(err.load(), err_msg);
  return GF_OK;
}

int gf_pair_hits_finish_device(const gf_index* idx, const void* d_hits, const void* d_totals, int64_t hits_cap,
                               const void* d_hit_bases, void* d_out, void* d_status, void* stream) {
  if (!idx || hits_cap < 0) return fail(GF_ERR_ARG, "null index or negative capacity");
  if (hits_cap == 0) return GF_OK;
  if (!d_hits || !d_totals || !d_hit_bases || !d_out || !d_status) return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((2 * hits_cap + 3) / 4, (int64_t)idx->n_cus * 8));
  hipLaunchKernelGGL(gf_k_pair_hits_finish, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const gf_pair_hit*)d_hits,
                     (const int64_t*)d_totals, hits_cap, (const uint8_t*)d_hit_bases, (const uint8_t*)idx->d_cat,
                     (const uint32_t*)idx->d_gene_off, (const uint32_t*)idx->d_gene_len, idx->table.n_genes,
                     (gf_readmatch*)d_out, (int32_t*)d_status);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_index_fusion_map_read(const gf_index* idx, const uint8_t* gene_reversed, const char* seq, int64_t len,
                             const gf_seqmatch* mapping, int32_t n_mapping, gf_readmatch* out) {
  if (!idx) return fail(GF_ERR_ARG, "null index");
  std::vector<const char*> p(idx->fusion_seq.size(), nullptr);
  std::vector<int64_t> l(idx->fusion_seq.size());
  for (size_t c = 0; c < p.size(); ++c) l[c] = (int64_t)idx->gene_len_h[c];
  for (int32_t k = 0; k < n_mapping && k < 2; ++k) {   // (make_match reads the genes of the two segments only)
    const int32_t c = mapping ? mapping[k].contig : -1;
    if (c >= 0 && (size_t)c < p.size()) {
      if (int rc = idx->ensure_fusion(c)) return rc;
      p[(size_t)c] = idx->fusion_seq[(size_t)c].data();
    }
  }
  return gf_fusion_map_read(p.data(), l.data(), (int32_t)p.size(), gene_reversed, seq, len, mapping, n_mapping, out);
}

// ---- FusionMapper::filter_matches (minus remove_alignables) and the order of sort_matches ----
static bool host_low_complexity(const char* s, int64_t n) {  // fusion_mapper.rs:559-569
  if (n < 20) return true;
  int diff = 0;  // dis_connected_count, utils/mod.rs:48-56
  for (int64_t i = 0; i + 1 < n; ++i) diff += s[i] != s[i + 1];
  return diff < 7;
}

int gf_readmatch_filter(const gf_readmatch* rm, const char* seq, int64_t len, int32_t deletion_threshold) {
  if (!rm || len < 0 || (len > 0 && !seq)) return fail(GF_ERR_ARG, "bad argument");
  const int64_t cut = (int64_t)rm->read_break + 1;
  if (cut < 0 || cut > len) return fail(GF_ERR_ARG, "read_break outside the read");  // the reference's subchars panics
  if (host_low_complexity(seq, cut) || host_low_complexity(seq + cut, len - cut)) return 1;
  if (rm->left_distance + rm->right_distance >= 5) return 2;
  if (rm->left_contig == rm->right_contig) {
    int64_t d = (int64_t)rm->left_position - (int64_t)rm->right_position;
    if (d < 0) d = -d;
    if (d < (int64_t)deletion_threshold) return 3;
  }
  return 0;
}

int gf_readmatch_order(int32_t a_break, int64_t a_len, const char* a_name, int64_t a_name_len, int32_t b_break,
                       int64_t b_len, const char* b_name, int64_t b_name_len) {
  if (a_break != b_break) return a_break > b_break ? -1 : 1;  // read_break descending
  if (a_len != b_len) return a_len < b_len ? -1 : 1;          // shorter read first
  const int64_t n = a_name_len < b_name_len ? a_name_len : b_name_len;
  int c = n > 0 ? memcmp(a_name, b_name, (size_t)n) : 0;
  if (c == 0) c = a_name_len < b_name_len ? -1 : (a_name_len > b_name_len ? 1 : 0);
  return c > 0 ? -1 : (c < 0 ? 1 : 0);  // name descending
}

// ---- SequenceReadPair::fast_merge on the device ----
// (d_l_qoff / d_r_qoff: where each read's qualities start in d_l_quals / d_r_quals — the FASTQ text, when the
//  qualities were left in it; null = at the bases' offsets)
static int merge_find_impl(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                           const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                           const void* d_r_offsets, int64_t n, int32_t max_read_len, void* d_out_len,
                           void* d_out_diff, void* stream, const void* d_l_qoff, const void* d_r_qoff) {
  if (!idx || n < 0 || max_read_len < 0) return fail(GF_ERR_ARG, "null index, negative n or max_read_len");
  if (n == 0) return GF_OK;
  if (!d_l_bases || !d_l_quals || !d_l_offsets || !d_r_bases || !d_r_quals || !d_r_offsets || !d_out_len ||
      !d_out_diff)
    return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  const uint8_t* lb = (const uint8_t*)d_l_bases; const uint8_t* lq = (const uint8_t*)d_l_quals;
  const uint8_t* rb = (const uint8_t*)d_r_bases; const uint8_t* rq = (const uint8_t*)d_r_quals;
  const int64_t* lo = (const int64_t*)d_l_offsets; const int64_t* ro = (const int64_t*)d_r_offsets;
  const int64_t* lqo = (const int64_t*)d_l_qoff; const int64_t* rqo = (const int64_t*)d_r_qoff;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)idx->n_cus * 32);
  if (max_read_len > 256) {  // beyond the packed kernels' word budget: the byte loop for every pair
    hipLaunchKernelGGL(gf_k_merge_find_bytes, dim3(grid), dim3(256), 0, st, lb, lq, lo, rb, rq, ro, n, lqo, rqo,
                       (int32_t*)d_out_len, (int32_t*)d_out_diff);
    GF_HIP(hipGetLastError());
    return GF_OK;
  }
  // groups of 64 pairs per wavefront, each packing its own spans into LDS
  const int g2 = (int)std::min<int64_t>((n + 255) / 256, (int64_t)idx->n_cus * 16);
  if (max_read_len <= 160)
    hipLaunchKernelGGL((gf_k_merge_find_stream<10>), dim3(g2), dim3(256), 0, st, lb, lq, lo, rb, rq, ro, n, lqo, rqo,
                       (int32_t*)d_out_len, (int32_t*)d_out_diff);
  else
    hipLaunchKernelGGL((gf_k_merge_find_stream<16>), dim3(g2), dim3(256), 0, st, lb, lq, lo, rb, rq, ro, n, lqo, rqo,
                       (int32_t*)d_out_len, (int32_t*)d_out_diff);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_fast_merge_find_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                              const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                              const void* d_r_offsets, int64_t n, int32_t max_read_len, void* d_out_len,
                              void* d_out_diff, void* stream) {
  return merge_find_impl(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n, max_read_len,
                         d_out_len, d_out_diff, stream, nullptr, nullptr);
}

static int merge_write_impl(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                            const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                            const void* d_r_offsets, int64_t n, const void* d_len, const void* d_out_pos,
                            void* d_out_bases, void* d_out_quals, void* stream, const void* d_l_qoff,
                            const void* d_r_qoff) {
  if (!idx || n < 0) return fail(GF_ERR_ARG, "null index or negative n");
  if (n == 0) return GF_OK;
  if (!d_l_bases || !d_l_quals || !d_l_offsets || !d_r_bases || !d_r_quals || !d_r_offsets || !d_len ||
      !d_out_pos || !d_out_bases)
    return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)idx->n_cus * 32);
  if (d_out_quals)
    hipLaunchKernelGGL(gf_k_merge_write, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_l_bases,
                       (const uint8_t*)d_l_quals, (const int64_t*)d_l_offsets, (const uint8_t*)d_r_bases,
                       (const uint8_t*)d_r_quals, (const int64_t*)d_r_offsets, n, (const int64_t*)d_l_qoff,
                       (const int64_t*)d_r_qoff, (const int32_t*)d_len,
                       (const int64_t*)d_out_pos, (uint8_t*)d_out_bases, (uint8_t*)d_out_quals);
  else  // the bases alone (gf_scan_pairs_device)
    hipLaunchKernelGGL(gf_k_merge_write_bases, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_l_bases,
                       (const uint8_t*)d_l_quals, (const int64_t*)d_l_offsets, (const uint8_t*)d_r_bases,
                       (const uint8_t*)d_r_quals, (const int64_t*)d_r_offsets, n, (const int64_t*)d_l_qoff,
                       (const int64_t*)d_r_qoff, (const int32_t*)d_len,
                       (const int64_t*)d_out_pos, (uint8_t*)d_out_bases);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_fast_merge_write_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                               const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                               const void* d_r_offsets, int64_t n, const void* d_len, const void* d_out_pos,
                               void* d_out_bases, void* d_out_quals, void* stream) {
  return merge_write_impl(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n, d_len, d_out_pos,
                          d_out_bases, d_out_quals, stream, nullptr, nullptr);
}

int gf_fast_merge_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals, const void* d_l_offsets,
                         const void* d_r_bases, const void* d_r_quals, const void* d_r_offsets, int64_t n,
                         int32_t max_read_len, const void* d_out_pos, void* d_out_bases, void* d_out_quals,
                         void* d_out_len, void* d_out_diff, void* stream) {
  int rc = gf_fast_merge_find_device(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n,
                                     max_read_len, d_out_len, d_out_diff, stream);
  if (rc != GF_OK) return rc;
  return gf_fast_merge_write_device(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n,
                                    d_out_len, d_out_pos, d_out_bases, d_out_quals, stream);
}

int gf_fast_merge(const gf_index* idx, const char* l_seq, const char* l_qual, int32_t len1, const char* r_seq,
                  const char* r_qual, int32_t len2, char* out_seq, char* out_qual, int32_t* out_len,
                  int32_t* out_diff) {
  if (!idx || len1 < 0 || len2 < 0 || !out_seq || !out_qual || !out_len || !out_diff)
    return fail(GF_ERR_ARG, "bad argument");
  DeviceGuard guard(idx->device);
  DevBuf<uint8_t> dl, dlq, dr, drq, dob, doq;
  DevBuf<int64_t> dlo, dro, dpos;
  DevBuf<int32_t> dlen, ddiff;
  // +64: the packing kernels read whole 16-byte chunks
  GF_HIP(dl.alloc((size_t)len1 + 64)); GF_HIP(dlq.alloc((size_t)len1 + 64));
  GF_HIP(dr.alloc((size_t)len2 + 64)); GF_HIP(drq.alloc((size_t)len2 + 64));
  GF_HIP(dob.alloc((size_t)len1 + len2 + 1)); GF_HIP(doq.alloc((size_t)len1 + len2 + 1));
  GF_HIP(dlo.alloc(2)); GF_HIP(dro.alloc(2)); GF_HIP(dpos.alloc(1)); GF_HIP(dlen.alloc(1)); GF_HIP(ddiff.alloc(1));
  const int64_t lo[2] = {0, len1}, ro[2] = {0, len2}, pos0 = 0;
  if (len1) { GF_HIP(hipMemcpy(dl.p, l_seq, (size_t)len1, hipMemcpyHostToDevice)); GF_HIP(hipMemcpy(dlq.p, l_qual, (size_t)len1, hipMemcpyHostToDevice)); }
  if (len2) { GF_HIP(hipMemcpy(dr.p, r_seq, (size_t)len2, hipMemcpyHostToDevice)); GF_HIP(hipMemcpy(drq.p, r_qual, (size_t)len2, hipMemcpyHostToDevice)); }
  GF_HIP(hipMemcpy(dlo.p, lo, sizeof lo, hipMemcpyHostToDevice));
  GF_HIP(hipMemcpy(dro.p, ro, sizeof ro, hipMemcpyHostToDevice));
  GF_HIP(hipMemcpy(dpos.p, &pos0, sizeof pos0, hipMemcpyHostToDevice));
  int rc = gf_fast_merge_device(idx, dl.p, dlq.p, dlo.p, dr.p, drq.p, dro.p, 1, std::max(len1, len2), dpos.p, dob.p,
                                doq.p, dlen.p, ddiff.p, nullptr);
  if (rc != GF_OK) return rc;
  GF_HIP(hipDeviceSynchronize());
  GF_HIP(hipMemcpy(out_len, dlen.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(out_diff, ddiff.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (*out_len <= 0) return 0;
  GF_HIP(hipMemcpy(out_seq, dob.p, (size_t)*out_len, hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(out_qual, doq.p, (size_t)*out_len, hipMemcpyDeviceToHost));
  return 1;
}

// ---- FastqReader::read on the device ----
// tiles of either kind a text of n_bytes can need: 16 KB text tiles, or tiles of 256 records (a
// record is at least its four newlines)
static inline int64_t fq_tiles(int64_t n_bytes) { return n_bytes / (4 * GF_FQ_RTILE) + 2; }

// the tile arrays, then (gf_fastq_index_device only) one newline bit per byte of the text
static inline int64_t fq_tile_arrays_bytes(int64_t n_bytes) {
  return (64 + fq_tiles(n_bytes) * (int64_t)(sizeof(uint32_t) + sizeof(int64_t)) + 16 + 15) & ~(int64_t)15;
}
static inline int64_t fq_text_tiles(int64_t n_bytes) { return (n_bytes + GF_FQ_TILE - 1) / GF_FQ_TILE; }

static inline int64_t fq_masks_bytes(int64_t n_bytes) { return fq_text_tiles(n_bytes) * (int64_t)(GF_CTHREADS * sizeof(uint64_t)); }
int64_t gf_fastq_workspace_bytes(int64_t n_bytes) {
  if (n_bytes < 0) return 0;
  // tile arrays | newline masks | the scan's round sums and bases (launch_scan_big)
  return fq_tile_arrays_bytes(n_bytes) + fq_masks_bytes(n_bytes) + (scan_rounds(fq_text_tiles(n_bytes)) + 2) * 16 + 32;
}

int gf_fastq_index_device(const gf_index* idx, const void* d_text, int64_t n_bytes, void* d_nl_pos, int64_t cap_lines,
                          void* d_n_lines, void* d_workspace, void* stream) {
  if (!idx || n_bytes < 0 || cap_lines < 0) return fail(GF_ERR_ARG, "null index or negative size");
  if (!d_n_lines || !d_workspace || (n_bytes > 0 && !d_text) || (cap_lines > 0 && !d_nl_pos))
    return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (n_bytes + GF_FQ_TILE - 1) / GF_FQ_TILE;
  uintptr_t w = ((uintptr_t)d_workspace + 15) & ~(uintptr_t)15;
  int64_t* tile_offsets = (int64_t*)w;
  uint32_t* tile_counts = (uint32_t*)(w + (size_t)fq_tiles(n_bytes) * sizeof(int64_t));
  uint64_t* masks = (uint64_t*)(w + (size_t)fq_tile_arrays_bytes(n_bytes));
  int64_t* n_lines = (int64_t*)d_n_lines;  // [0] lines, [1] newlines
  if (ntiles > 0) {
    hipLaunchKernelGGL(gf_k_fq_count, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, (const uint8_t*)d_text, n_bytes,
                       tile_counts, masks);
    GF_HIP(hipGetLastError());
  }
  {
    int64_t* round_offsets = (int64_t*)(((uintptr_t)masks + (size_t)fq_masks_bytes(n_bytes) + 15) & ~(uintptr_t)15);
    uint32_t* round_sums = (uint32_t*)(round_offsets + scan_rounds(ntiles) + 1);
    launch_scan_big(st, ntiles, tile_counts, tile_offsets, n_lines + 1, round_sums, round_offsets);  // (<= 16 384 newlines per tile)
  }
  GF_HIP(hipGetLastError());
  if (ntiles > 0) {
    hipLaunchKernelGGL(gf_k_fq_write, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, (const uint8_t*)d_text, n_bytes,
                       tile_offsets, n_lines + 1, (const uint64_t*)masks, (int64_t*)d_nl_pos, cap_lines, n_lines);
    GF_HIP(hipGetLastError());
  } else {
    GF_HIP(hipMemsetAsync(n_lines, 0, sizeof(int64_t), st));
  }
  return GF_OK;
}

// (d_qual_off != null: the lean form — no qualities copied, d_qual_off[r] = where record r's quality line starts)
static int fastq_gather_impl(const gf_index* idx, const void* d_text, int64_t n_bytes, const void* d_nl_pos,
                             int64_t n_newlines, int64_t n_records, void* d_offsets, void* d_bases, void* d_quals,
                             int64_t cap_bytes, void* d_n_bad, void* d_workspace, void* stream, void* d_qual_off) {
  if (!idx || n_bytes < 0 || n_newlines < 0 || n_records < 0 || cap_bytes < 0)
    return fail(GF_ERR_ARG, "null index or negative size");
  if (!d_offsets || !d_n_bad || !d_workspace) return fail(GF_ERR_ARG, "null device pointer");
  // record i owns lines 4i .. 4i+3: all of them must exist
  if (4 * n_records > n_newlines + 1) return fail(GF_ERR_ARG, "n_records exceeds the lines of the text");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  GF_HIP(hipMemsetAsync(d_n_bad, 0, sizeof(unsigned long long), st));
  if (n_records == 0) {
    GF_HIP(hipMemsetAsync(d_offsets, 0, sizeof(int64_t), st));
    return GF_OK;
  }
  if (!d_text || !d_nl_pos || (cap_bytes > 0 && (!d_bases || (!d_quals && !d_qual_off)))) return fail(GF_ERR_ARG, "null device pointer");
  const int64_t ntiles = (n_records + GF_FQ_RTILE - 1) / GF_FQ_RTILE;
  if (ntiles >= fq_tiles(n_bytes)) return fail(GF_ERR_ARG, "n_records impossible for a text of n_bytes");
  uintptr_t w = ((uintptr_t)d_workspace + 15) & ~(uintptr_t)15;
  int64_t* tile_offsets = (int64_t*)w;
  uint32_t* tile_counts = (uint32_t*)(w + (size_t)fq_tiles(n_bytes) * sizeof(int64_t));
  int64_t* total = tile_offsets + fq_tiles(n_bytes) - 1;  // last slot is spare
  hipLaunchKernelGGL(gf_k_fq_lens, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, (const int64_t*)d_nl_pos, n_newlines,
                     n_bytes, n_records, tile_counts);
  launch_scan(st, ntiles, tile_counts, tile_offsets, total);
  if (d_qual_off)
    hipLaunchKernelGGL((gf_k_fq_gather<false>), dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, (const uint8_t*)d_text,
                       (const int64_t*)d_nl_pos, n_newlines, n_bytes, n_records, tile_offsets, (int64_t*)d_offsets,
                       (uint8_t*)d_bases, (uint8_t*)nullptr, cap_bytes, (unsigned long long*)d_n_bad, (int64_t*)d_qual_off);
  else
    hipLaunchKernelGGL((gf_k_fq_gather<true>), dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, (const uint8_t*)d_text,
                       (const int64_t*)d_nl_pos, n_newlines, n_bytes, n_records, tile_offsets, (int64_t*)d_offsets,
                       (uint8_t*)d_bases, (uint8_t*)d_quals, cap_bytes, (unsigned long long*)d_n_bad, (int64_t*)nullptr);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_fastq_gather_device(const gf_index* idx, const void* d_text, int64_t n_bytes, const void* d_nl_pos,
                           int64_t n_newlines, int64_t n_records, void* d_offsets, void* d_bases, void* d_quals,
                           int64_t cap_bytes, void* d_n_bad, void* d_workspace, void* stream) {
  return fastq_gather_impl(idx, d_text, n_bytes, d_nl_pos, n_newlines, n_records, d_offsets, d_bases, d_quals, cap_bytes,
                           d_n_bad, d_workspace, stream, nullptr);
}

int gf_fastq_gather_lean_device(const gf_index* idx, const void* d_text, int64_t n_bytes, const void* d_nl_pos,
                                int64_t n_newlines, int64_t n_records, void* d_offsets, void* d_bases, int64_t cap_bytes,
                                void* d_qual_off, void* d_n_bad, void* d_workspace, void* stream) {
  if (!d_qual_off) return fail(GF_ERR_ARG, "null quality offsets");
  return fastq_gather_impl(idx, d_text, n_bytes, d_nl_pos, n_newlines, n_records, d_offsets, d_bases, nullptr, cap_bytes,
                           d_n_bad, d_workspace, stream, d_qual_off);
}

// ---- the pair policy of scan_pair_end for a whole pack, device resident (gf_pair_kernels.h) ----
int gf_index_set_gene_reversed(gf_index* idx, const uint8_t* gene_reversed, int32_t n_genes) {
  if (!idx) return fail(GF_ERR_ARG, "null index");
  if (n_genes != idx->table.n_genes) return fail(GF_ERR_ARG, "n_genes differs from the index");
  if (n_genes > 0 && !gene_reversed) return fail(GF_ERR_ARG, "null flags");
  DeviceGuard guard(idx->device);
  if (!idx->d_gene_rev) GF_HIP(hipMalloc((void**)&idx->d_gene_rev, (size_t)std::max(n_genes, 1)));
  if (n_genes > 0) GF_HIP(hipMemcpy(idx->d_gene_rev, gene_reversed, (size_t)n_genes, hipMemcpyHostToDevice));
  return GF_OK;
}

int64_t gf_scan_pairs_retry_capacity(int64_t n) { return n < 0 ? 0 : std::max<int64_t>(4096, n / 4); }

static int scan_pairs_impl(const gf_index* idx, const void* d_l_bases, const void* d_l_quals, const void* d_l_offsets,
                           int64_t l_bytes, const void* d_r_bases, const void* d_r_quals, const void* d_r_offsets,
                           int64_t r_bytes, int64_t n, int32_t max_read_len, int64_t pair_id_base, int64_t retry_cap,
                           void* d_hits, int64_t hits_cap, void* d_hit_bases, void* d_hit_quals, int64_t hit_bytes_cap,
                           void* d_totals, void* stream, const void* d_l_qoff, const void* d_r_qoff) {
  if (!idx || n < 0 || l_bytes < 0 || r_bytes < 0 || hits_cap < 0 || hit_bytes_cap < 0 || max_read_len < 0)
    return fail(GF_ERR_ARG, "null index or negative size");
  if (!d_totals) return fail(GF_ERR_ARG, "null totals");
  if (n > 0 && (!d_l_bases || !d_l_quals || !d_l_offsets || !d_r_bases || !d_r_quals || !d_r_offsets))
    return fail(GF_ERR_ARG, "null device pointer");
  if ((hits_cap > 0 && !d_hits) || (hit_bytes_cap > 0 && (!d_hit_bases || !d_hit_quals)))
    return fail(GF_ERR_ARG, "null output pointer");
  if (n > (int64_t)0x7FFFFFFF / 3) return fail(GF_ERR_CAPACITY, "more than 2^31 / 3 pairs in one pack");
  const int64_t merged_max = 2 * (int64_t)max_read_len;  // a merged read is shorter than len1 + len2
  if (merged_max > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "2 * max_read_len exceeds GF_MAX_READ_LEN");
  DeviceGuard guard(idx->device);
  hipStream_t st = (hipStream_t)stream;
  GF_HIP(hipMemsetAsync(d_totals, 0, 8 * sizeof(int64_t), st));
  if (n == 0) return GF_OK;
  if (retry_cap <= 0) retry_cap = gf_scan_pairs_retry_capacity(n);
  retry_cap = std::min<int64_t>(retry_cap, 3 * n);
  const int64_t retry_bytes_cap = std::min<int64_t>(retry_cap * merged_max, l_bytes + r_bytes + 64);
  const int64_t ntiles = (n + GF_PTILE - 1) / GF_PTILE;
  const int64_t nctiles = (n + GF_CTILE - 1) / GF_CTILE;

  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
  } cv;
  (void)al;
  const size_t o_mlen = cv.take((size_t)n * 4), o_mdiff = cv.take((size_t)n * 4), o_moff = cv.take(((size_t)n + 1) * 8);
  const size_t o_mrank = cv.take((size_t)n * 4), o_coff = cv.take(((size_t)n + 1) * 8);
  const size_t o_mb = cv.take((size_t)(l_bytes + r_bytes) + 64);
  const size_t o_cM = cv.take((size_t)n), o_c1 = cv.take((size_t)n), o_c2 = cv.take((size_t)n);
  const size_t o_mM = cv.take((size_t)n * 32), o_m1 = cv.take((size_t)n * 32), o_m2 = cv.take((size_t)n * 32);
  const size_t o_st = cv.take((size_t)n * 3), o_slot = cv.take((size_t)n * 3 * 4);
  const size_t o_tc = cv.take((size_t)std::max(ntiles, nctiles) * 4 * 2);      // two uint32 per tile
  const size_t o_to = cv.take((size_t)std::max(ntiles, nctiles) * 8 * 2);      // two int64 per tile
  const size_t o_scal = cv.take(256);                                          // scalars: totals of the scans, merged pairs
  const size_t o_roff = cv.take(((size_t)retry_cap + 1) * 8);
  const size_t o_rb = cv.take((size_t)retry_bytes_cap + 64), o_rq = cv.take((size_t)retry_bytes_cap + 64);
  const size_t o_cR = cv.take((size_t)retry_cap), o_mR = cv.take((size_t)retry_cap * 32);

  const std::shared_ptr<Workspace> pwse = ws_pool(true).entry(idx->device, st);
  std::lock_guard<std::mutex> lk(pwse->mu);  // held until this call's launches are queued
  void* base = nullptr;
  int rc = acquire_workspace(*pwse, st, cv.off, &base);
  if (rc != GF_OK) return rc;
  uint8_t* wp = (uint8_t*)base;
  int32_t* m_len = (int32_t*)(wp + o_mlen); int32_t* m_diff = (int32_t*)(wp + o_mdiff); int64_t* m_off = (int64_t*)(wp + o_moff);
  int32_t* m_rank = (int32_t*)(wp + o_mrank); int64_t* c_off = (int64_t*)(wp + o_coff);
  uint8_t* mb = wp + o_mb;
  uint8_t *cM = wp + o_cM, *c1 = wp + o_c1, *c2 = wp + o_c2;
  gf_seqmatch *mM = (gf_seqmatch*)(wp + o_mM), *m1 = (gf_seqmatch*)(wp + o_m1), *m2 = (gf_seqmatch*)(wp + o_m2);
  uint8_t* stt = wp + o_st; int32_t* slot_of = (int32_t*)(wp + o_slot);
  const int64_t tmax = std::max(ntiles, nctiles);
  uint32_t *tcA = (uint32_t*)(wp + o_tc), *tcB = tcA + tmax;
  int64_t *toA = (int64_t*)(wp + o_to), *toB = toA + tmax;
  int64_t* scal = (int64_t*)(wp + o_scal);  // [0] merged bytes, [1] retries, [2] retry bytes, [3] hits, [4] hit bytes, [5] merged pairs
  int64_t* r_off = (int64_t*)(wp + o_roff); uint8_t* rb = wp + o_rb; uint8_t* rq = wp + o_rq;
  uint8_t* cR = wp + o_cR; gf_seqmatch* mR = (gf_seqmatch*)(wp + o_mR);
  int64_t* totals = (int64_t*)d_totals;
  GF_HIP(hipMemsetAsync(scal, 0, 256, st));

  // 1. fast_merge: lengths, then the slots of the merged reads (exclusive scan), then the reads
  rc = merge_find_impl(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n, max_read_len,
                       m_len, m_diff, stream, d_l_qoff, d_r_qoff);
  if (rc != GF_OK) return rc;
  // scalars: [0] merged bytes, [1] retries, [2] retry bytes, [3] hits, [4] hit bytes, [5] merged pairs
  hipLaunchKernelGGL(gf_k_len_tile_sums, dim3((unsigned)nctiles), dim3(GF_CTHREADS), 0, st, (const int32_t*)m_len, n, tcA, tcB);
  launch_scan(st, nctiles, (const uint32_t*)tcA, toA, scal + 0, (const uint32_t*)tcB, toB, scal + 5);
  hipLaunchKernelGGL(gf_k_len_offsets, dim3((unsigned)nctiles), dim3(GF_CTHREADS), 0, st, (const int32_t*)m_len, n,
                     (const int64_t*)toA, (const int64_t*)toB, (const int64_t*)(scal + 0), m_off, m_rank, c_off);
  hipLaunchKernelGGL(gf_k_len_tail, dim3((unsigned)std::min<int64_t>((n + 256) / 256, 2048)), dim3(256), 0, st,
                     (const int64_t*)(scal + 5), (const int64_t*)(scal + 0), n, c_off);
  GF_HIP(hipGetLastError());
  rc = merge_write_impl(idx, d_l_bases, d_l_quals, d_l_offsets, d_r_bases, d_r_quals, d_r_offsets, n, m_len, m_off,
                        mb, nullptr, stream, d_l_qoff, d_r_qoff);
  if (rc != GF_OK) return rc;
  // 2. the merged reads; R1 and R2 of the pairs that did not merge (in place, the others skipped)
  rc = map_reads_device_impl(idx, mb, c_off, n, (int32_t)std::max<int64_t>(merged_max, 1), cM, mM, stream, nullptr, nullptr,
                             nullptr, 0, (const int64_t*)(scal + 5));  // (the merged reads are the first scal[5] of the n slots)
  if (rc != GF_OK) return rc;
  rc = map_reads_device_impl(idx, d_l_bases, d_l_offsets, n, std::max(max_read_len, 1), c1, m1, stream, m_len);
  if (rc != GF_OK) return rc;
  rc = map_reads_device_impl(idx, d_r_bases, d_r_offsets, n, std::max(max_read_len, 1), c2, m2, stream, m_len);
  if (rc != GF_OK) return rc;
  // 3. matches as they are / reverse-complement retries
  GfPairIn P;
  P.l_bases = (const uint8_t*)d_l_bases; P.l_quals = (const uint8_t*)d_l_quals; P.l_off = (const int64_t*)d_l_offsets;
  P.r_bases = (const uint8_t*)d_r_bases; P.r_quals = (const uint8_t*)d_r_quals; P.r_off = (const int64_t*)d_r_offsets;
  P.l_qoff = (const int64_t*)d_l_qoff; P.r_qoff = (const int64_t*)d_r_qoff;
  P.m_bases = mb; P.m_off = m_off; P.m_len = m_len; P.m_diff = m_diff; P.m_rank = m_rank;
  P.cM = cM; P.c1 = c1; P.c2 = c2; P.mM = mM; P.m1 = m1; P.m2 = m2;
  P.gene_reversed = idx->d_gene_rev; P.n_genes = idx->table.n_genes;
  hipLaunchKernelGGL(gf_k_pair_classify, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, P, n, stt, tcA, tcB);
  launch_scan(st, ntiles, (const uint32_t*)tcA, toA, scal + 1, (const uint32_t*)tcB, toB, scal + 2);
  hipLaunchKernelGGL(gf_k_pair_retry_write, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, P, n, (const uint8_t*)stt,
                     (const int64_t*)toA, (const int64_t*)toB, retry_cap, retry_bytes_cap, r_off, rb, rq, slot_of);
  hipLaunchKernelGGL(gf_k_pair_retry_tail, dim3((unsigned)std::min<int64_t>((retry_cap + 256) / 256, 1024)), dim3(256), 0, st,
                     (const int64_t*)(scal + 1), (const int64_t*)(scal + 2), retry_cap, retry_bytes_cap, r_off, totals,
                     (unsigned int*)(scal + 6));
  GF_HIP(hipGetLastError());
  // The retries — a few per ten thousand pairs, every one a read that does map — go straight to the exact
  // wave-per-read kernel, as many as there are (the count is on the device).  Through the flat pipeline like the other
  // passes (GF_RETRY_FLAT=1) they cost 0.41 ms per 10 M pairs: launches over retry_cap mostly empty slots, and a
  // bucket kernel whose every lane walks a junction read's thirty windows one probe after the other.
  static const bool retry_flat = getenv("GF_RETRY_FLAT") != nullptr;
  if (retry_flat) {
    rc = map_reads_device_impl(idx, rb, r_off, retry_cap, (int32_t)std::max<int64_t>(merged_max, 1), cR, mR, stream, nullptr);
    if (rc != GF_OK) return rc;
  } else {
    GfTable T = idx->table;
    T.skip = nullptr;
    T.fixed_len = 0;
    T.n_dev = nullptr;
    const unsigned int* n_exact = (const unsigned int*)(scal + 6);
    if (merged_max <= 256)
      hipLaunchKernelGGL((gf_k_map_reads_list<256, 4, false>), dim3(idx->n_cus * 8), dim3(256), 0, st, T, (const uint8_t*)rb,
                         (const uint32_t*)nullptr, (const uint16_t*)nullptr, (const int64_t*)r_off, (const uint32_t*)nullptr,
                         (int64_t)1, n_exact, cR, mR);
    else if (merged_max <= 1024)
      hipLaunchKernelGGL((gf_k_map_reads_list<1024, 4, false>), dim3(idx->n_cus * 3), dim3(256), 0, st, T, (const uint8_t*)rb,
                         (const uint32_t*)nullptr, (const uint16_t*)nullptr, (const int64_t*)r_off, (const uint32_t*)nullptr,
                         (int64_t)1, n_exact, cR, mR);
    else
      hipLaunchKernelGGL((gf_k_map_reads_list<4096, 2, false>), dim3(idx->n_cus * 4), dim3(128), 0, st, T, (const uint8_t*)rb,
                         (const uint32_t*)nullptr, (const uint16_t*)nullptr, (const int64_t*)r_off, (const uint32_t*)nullptr,
                         (int64_t)1, n_exact, cR, mR);
    GF_HIP(hipGetLastError());
  }
  // 4. the matches, in push order, with their reads
  hipLaunchKernelGGL(gf_k_pair_final_count, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, P, n, (const uint8_t*)stt,
                     (const int32_t*)slot_of, (const uint8_t*)cR, (const gf_seqmatch*)mR, tcA, tcB);
  launch_scan(st, ntiles, (const uint32_t*)tcA, toA, scal + 3, (const uint32_t*)tcB, toB, scal + 4);
  hipLaunchKernelGGL(gf_k_pair_final_write, dim3((unsigned)ntiles), dim3(GF_CTHREADS), 0, st, P, n, pair_id_base,
                     (const uint8_t*)stt, (const int32_t*)slot_of, (const uint8_t*)cR, (const gf_seqmatch*)mR,
                     (const int64_t*)r_off, (const uint8_t*)rb, (const uint8_t*)rq, (const int64_t*)toA, (const int64_t*)toB,
                     (gf_pair_hit*)d_hits, hits_cap, (uint8_t*)d_hit_bases, (uint8_t*)d_hit_quals, hit_bytes_cap);
  hipLaunchKernelGGL(gf_k_pair_totals, dim3(1), dim3(1), 0, st, (const int64_t*)(scal + 3), (const int64_t*)(scal + 4),
                     (const int64_t*)(scal + 5), hits_cap, hit_bytes_cap, totals);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_scan_pairs_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals, const void* d_l_offsets,
                         int64_t l_bytes, const void* d_r_bases, const void* d_r_quals, const void* d_r_offsets,
                         int64_t r_bytes, int64_t n, int32_t max_read_len, int64_t pair_id_base, int64_t retry_cap,
                         void* d_hits, int64_t hits_cap, void* d_hit_bases, void* d_hit_quals, int64_t hit_bytes_cap,
                         void* d_totals, void* stream) {
  return scan_pairs_impl(idx, d_l_bases, d_l_quals, d_l_offsets, l_bytes, d_r_bases, d_r_quals, d_r_offsets, r_bytes, n,
                         max_read_len, pair_id_base, retry_cap, d_hits, hits_cap, d_hit_bases, d_hit_quals, hit_bytes_cap,
                         d_totals, stream, nullptr, nullptr);
}

int gf_scan_pairs_text_device(const gf_index* idx, const void* d_l_bases, const void* d_l_text, const void* d_l_qual_off,
                              const void* d_l_offsets, int64_t l_bytes, const void* d_r_bases, const void* d_r_text,
                              const void* d_r_qual_off, const void* d_r_offsets, int64_t r_bytes, int64_t n,
                              int32_t max_read_len, int64_t pair_id_base, int64_t retry_cap, void* d_hits, int64_t hits_cap,
                              void* d_hit_bases, void* d_hit_quals, int64_t hit_bytes_cap, void* d_totals, void* stream) {
  if (n > 0 && (!d_l_qual_off || !d_r_qual_off)) return fail(GF_ERR_ARG, "null quality offsets");
  return scan_pairs_impl(idx, d_l_bases, d_l_text, d_l_offsets, l_bytes, d_r_bases, d_r_text, d_r_offsets, r_bytes, n,
                         max_read_len, pair_id_base, retry_cap, d_hits, hits_cap, d_hit_bases, d_hit_quals, hit_bytes_cap,
                         d_totals, stream, d_l_qual_off, d_r_qual_off);
}

// ---- streaming host entry: packs submitted ahead of the ones being mapped ---------------------
// The consumer loop of the reference takes packs of reads off a queue while the producer keeps
// reading the FASTQ (pescanner.rs:255-311).  Here a pack is SUBMITTED (copy in, mapping, compaction,
// hit records out: all queued on the slot's own stream) and COLLECTED later; with `depth` slots the
// copy of pack k+1 crosses the link while the kernels of pack k run and the hits of pack k-1 go back.
struct gf_stream {
  gf_index* ix = nullptr;
  int depth = 0;
  int64_t max_reads = 0, max_bytes = 0, pin_cap = 0;
  struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t done{};
    uint8_t* arena = nullptr;  // device: bases | offsets | counts | matches | hits | total | compaction workspace
    uint8_t* d_bases = nullptr; int64_t* d_off = nullptr; uint8_t* d_counts = nullptr; gf_seqmatch* d_matches = nullptr;
    gf_hit* d_hits = nullptr; int64_t* d_total = nullptr; uint8_t* d_cws = nullptr;
    gf_hit* h_hits = nullptr;  // pinned: the first pin_cap records of the pack
    int64_t* h_total = nullptr;
    bool inflight = false;
    int64_t n = 0;
    ZeroCopy zc;              // a pack of up to GF_PACK_CALL_READS reads takes the zero-copy route (one launch, no copies)
    bool zc_call = false;
    int64_t read_id_base = 0;
  };
  std::vector<Slot> slots;
  int head = 0, tail = 0, live = 0;
  bool counted = false;  // registered in ix->open_streams
};

int gf_copy_from_host_device(const gf_index* idx, const void* h_src, void* d_dst, int64_t nbytes, void* stream) {
  if (!idx || nbytes < 0) return fail(GF_ERR_ARG, "null index or negative size");
  if (nbytes == 0) return GF_OK;
  if (!h_src || !d_dst) return fail(GF_ERR_ARG, "null pointer");
  DeviceGuard guard(idx->device);
  GF_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return GF_OK;
}

void* gf_host_alloc(int64_t bytes) {
  if (bytes <= 0) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    g_err = "hipHostMalloc failed";
    return nullptr;
  }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess && d) {  // known to the zero-copy calls: they read it in place
    std::unique_lock<std::shared_mutex> lk(g_pinned_mu);
    g_pinned.push_back({(const char*)p, (size_t)bytes, (const uint8_t*)d});
  } else {
    (void)hipGetLastError();
  }
  return p;
}

void gf_host_free(void* p) {
  if (!p) return;
  {
    std::unique_lock<std::shared_mutex> lk(g_pinned_mu);
    for (size_t i = 0; i < g_pinned.size(); ++i)
      if (g_pinned[i].host == (const char*)p) {
        g_pinned.erase(g_pinned.begin() + (long)i);
        break;
      }
  }
  (void)hipHostFree(p);
}

void gf_stream_close(gf_stream* s) {
  if (!s) return;
  DeviceGuard guard(s->ix->device);
  if (s->counted) s->ix->open_streams.fetch_sub(1);
  for (auto& sl : s->slots) {
    if (sl.st) (void)hipStreamSynchronize(sl.st);
    if (sl.arena) (void)hipFree(sl.arena);
    if (sl.h_hits) (void)hipHostFree(sl.h_hits);
    if (sl.h_total) (void)hipHostFree(sl.h_total);
    sl.zc.release();
    if (sl.st) {
      (void)hipEventDestroy(sl.done);
      // the mapping workspace cached for this stream goes with it
      ws_pool(false).drop(s->ix->device, sl.st);
      (void)hipStreamDestroy(sl.st);
    }
  }
  delete s;
}

int gf_stream_open(const gf_index* idx, int64_t max_reads, int64_t max_bytes, int32_t depth, gf_stream** out) {
  if (!out) return fail(GF_ERR_ARG, "out is null");
  *out = nullptr;
  if (!idx || max_reads <= 0 || max_bytes < 0 || depth < 1 || depth > 16) return fail(GF_ERR_ARG, "bad argument");
  DeviceGuard guard(idx->device);
  std::unique_ptr<gf_stream, void (*)(gf_stream*)> s(new gf_stream(), gf_stream_close);
  s->ix = const_cast<gf_index*>(idx);
  s->ix->open_streams.fetch_add(1);
  s->counted = true;
  s->depth = depth;
  s->max_reads = max_reads;
  s->max_bytes = max_bytes;
  s->pin_cap = std::max<int64_t>(4096, max_reads / 16);
  s->slots.resize((size_t)depth);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t sz_b = al((size_t)max_bytes + 64), sz_o = al(((size_t)max_reads + 1) * 8), sz_c = al((size_t)max_reads + 1);
  const size_t sz_m = al(((size_t)max_reads * 2 + 1) * sizeof(gf_seqmatch)), sz_h = al(((size_t)max_reads + 1) * sizeof(gf_hit));
  const size_t sz_w = al((size_t)gf_compact_workspace_bytes(max_reads) + 16);
  for (auto& sl : s->slots) {
    GF_HIP(hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
    GF_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    GF_HIP(hipMalloc((void**)&sl.arena, sz_b + sz_o + sz_c + sz_m + sz_h + 256 + sz_w));
    uint8_t* wp = sl.arena;
    sl.d_bases = wp; wp += sz_b;
    sl.d_off = (int64_t*)wp; wp += sz_o;
    sl.d_counts = wp; wp += sz_c;
    sl.d_matches = (gf_seqmatch*)wp; wp += sz_m;
    sl.d_hits = (gf_hit*)wp; wp += sz_h;
    sl.d_total = (int64_t*)wp; wp += 256;
    sl.d_cws = wp;
    GF_HIP(hipHostMalloc((void**)&sl.h_hits, (size_t)s->pin_cap * sizeof(gf_hit), hipHostMallocDefault));
    GF_HIP(hipHostMalloc((void**)&sl.h_total, 64, hipHostMallocDefault));
  }
  *out = s.release();
  return GF_OK;
}

int gf_stream_submit(gf_stream* s, const char* bases, const int64_t* offsets, int64_t n, int64_t read_id_base) {
  if (!s || n < 0 || (n > 0 && !offsets)) return fail(GF_ERR_ARG, "bad argument");
  if (n > s->max_reads) return fail(GF_ERR_CAPACITY, "pack has more reads than the stream was opened for");
  if (s->live == s->depth) return fail(GF_ERR_CAPACITY, "every slot is in flight: collect a pack first");
  int64_t maxlen = 0;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t l = offsets[r + 1] - offsets[r];
    if (l < 0) return fail(GF_ERR_ARG, "offsets must be non-decreasing");
    maxlen = std::max(maxlen, l);
  }
  if (maxlen > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "a read exceeds GF_MAX_READ_LEN");
  const int64_t b0 = n > 0 ? offsets[0] : 0, b1 = n > 0 ? offsets[n] : 0;
  if (b1 - b0 > s->max_bytes) return fail(GF_ERR_CAPACITY, "pack has more bytes than the stream was opened for");
  if (b1 > b0 && !bases) return fail(GF_ERR_ARG, "bases is null");
  DeviceGuard guard(s->ix->device);
  gf_stream::Slot& sl = s->slots[(size_t)s->head];
  sl.n = n;
  sl.zc_call = false;
  sl.read_id_base = read_id_base;
  if (zero_copy_call(s->ix, n)) {  // a pack of the reference's size: see ZeroCopy
    const int rc = zc_submit(s->ix, sl.zc, sl.st, bases, offsets, n);
    if (rc != GF_OK) return rc;
    sl.zc_call = true;
    sl.inflight = true;
    s->head = (s->head + 1) % s->depth;
    s->live += 1;
    return GF_OK;
  }
  if (n > 0) {
    const size_t lead = (size_t)((uintptr_t)(bases + b0) & 15u);  // keep the span's 16-byte phase
    struct Drain { hipStream_t st; bool armed = true; ~Drain() { if (armed) (void)hipStreamSynchronize(st); } } drain{sl.st};  // (see gf_stream_submit_packed)
    if (b1 > b0) GF_HIP(hipMemcpyAsync(sl.d_bases + lead, bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, sl.st));
    GF_HIP(hipMemcpyAsync(sl.d_off, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, sl.st));
    int rc = gf_map_reads_device(s->ix, sl.d_bases + lead - b0, sl.d_off, n, (int32_t)std::max<int64_t>(maxlen, 1),
                                 sl.d_counts, sl.d_matches, (void*)sl.st);
    if (rc != GF_OK) return rc;
    rc = gf_compact_hits_device(s->ix, sl.d_counts, sl.d_matches, n, read_id_base, sl.d_hits, n, sl.d_total, sl.d_cws,
                                (void*)sl.st);
    if (rc != GF_OK) return rc;
    GF_HIP(hipMemcpyAsync(sl.h_total, sl.d_total, 8, hipMemcpyDeviceToHost, sl.st));
    GF_HIP(hipMemcpyAsync(sl.h_hits, sl.d_hits, (size_t)std::min(n, s->pin_cap) * sizeof(gf_hit), hipMemcpyDeviceToHost,
                          sl.st));
    drain.armed = false;
  } else {
    *sl.h_total = 0;
  }
  GF_HIP(hipEventRecord(sl.done, sl.st));
  sl.inflight = true;
  s->head = (s->head + 1) % s->depth;
  s->live += 1;
  return GF_OK;
}

int gf_stream_submit_packed(gf_stream* s, const uint32_t* pk, const uint16_t* iv, const int64_t* offsets, int64_t n,
                            int64_t read_id_base) {
  if (!s || n < 0 || (n > 0 && !offsets)) return fail(GF_ERR_ARG, "bad argument");
  if (n > s->max_reads) return fail(GF_ERR_CAPACITY, "pack has more reads than the stream was opened for");
  if (s->live == s->depth) return fail(GF_ERR_CAPACITY, "every slot is in flight: collect a pack first");
  int64_t maxlen = 0;
  for (int64_t r = 0; r < n; ++r) {
    const int64_t l = offsets[r + 1] - offsets[r];
    if (l < 0) return fail(GF_ERR_ARG, "offsets must be non-decreasing");
    maxlen = std::max(maxlen, l);
  }
  if (maxlen > GF_MAX_READ_LEN) return fail(GF_ERR_READ_TOO_LONG, "a read exceeds GF_MAX_READ_LEN");
  const int64_t b0 = n > 0 ? offsets[0] : 0, b1 = n > 0 ? offsets[n] : 0;
  if (b0 < 0) return fail(GF_ERR_ARG, "negative offset");
  if (b1 - b0 > s->max_bytes) return fail(GF_ERR_CAPACITY, "pack has more bases than the stream was opened for");
  if (b1 > b0 && (!pk || !iv)) return fail(GF_ERR_ARG, "null packed stream");
  DeviceGuard guard(s->ix->device);
  gf_stream::Slot& sl = s->slots[(size_t)s->head];
  sl.n = n;
  if (n > 0) {
    // chunks c0 .. c1-1 cover the pack's bases and the four chunks past them that every packed buffer carries
    // (gf_packed_chunks: the kernels' staging reads past a tile's last chunk; the host's arrays hold them — the next
    // reads' chunks inside a buffer, gf_pack_bases_host's padding at its end); the slot's base area
    // (max_bytes + 64 bytes) holds them: 6 bytes per chunk of 16 bases
    const int64_t c0 = b0 >> 4, c1 = ((b1 + 15) >> 4) + 4, nc = c1 - c0;
    uint32_t* d_pk = (uint32_t*)sl.d_bases;
    uint16_t* d_iv = (uint16_t*)(sl.d_bases + (((size_t)nc * 4 + 15) & ~(size_t)15));
    if ((((size_t)nc * 4 + 15) & ~(size_t)15) + (size_t)nc * 2 > (size_t)s->max_bytes + 64)
      return fail(GF_ERR_CAPACITY, "pack has more bases than the stream was opened for");
    // whatever way this call ends, the copies out of the caller's pinned pk / iv / offsets are not left pending on
    // an error return (the caller may reuse them at once: the pack never became a slot in flight)
    struct Drain { hipStream_t st; bool armed = true; ~Drain() { if (armed) (void)hipStreamSynchronize(st); } } drain{sl.st};
    GF_HIP(hipMemcpyAsync(d_pk, pk + c0, (size_t)nc * 4, hipMemcpyHostToDevice, sl.st));
    GF_HIP(hipMemcpyAsync(d_iv, iv + c0, (size_t)nc * 2, hipMemcpyHostToDevice, sl.st));
    GF_HIP(hipMemcpyAsync(sl.d_off, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, sl.st));
    // the device arrays start at chunk c0 of the host's stream: shift the pointers, the offsets stay as they are
    int rc = gf_map_reads_packed_device(s->ix, d_pk - c0, d_iv - c0, sl.d_off, n, (int32_t)std::max<int64_t>(maxlen, 1),
                                        sl.d_counts, sl.d_matches, (void*)sl.st);
    if (rc != GF_OK) return rc;
    rc = gf_compact_hits_device(s->ix, sl.d_counts, sl.d_matches, n, read_id_base, sl.d_hits, n, sl.d_total, sl.d_cws,
                                (void*)sl.st);
    if (rc != GF_OK) return rc;
    GF_HIP(hipMemcpyAsync(sl.h_total, sl.d_total, 8, hipMemcpyDeviceToHost, sl.st));
    GF_HIP(hipMemcpyAsync(sl.h_hits, sl.d_hits, (size_t)std::min(n, s->pin_cap) * sizeof(gf_hit), hipMemcpyDeviceToHost,
                          sl.st));
    drain.armed = false;  // the pack is in flight: gf_stream_collect waits for it
  } else {
    *sl.h_total = 0;
  }
  GF_HIP(hipEventRecord(sl.done, sl.st));
  sl.inflight = true;
  s->head = (s->head + 1) % s->depth;
  s->live += 1;
  return GF_OK;
}

int gf_stream_collect(gf_stream* s, gf_hit* out_hits, int64_t cap, int64_t* out_n) {
  if (!s || !out_n || cap < 0 || (cap > 0 && !out_hits)) return fail(GF_ERR_ARG, "bad argument");
  *out_n = 0;
  if (s->live == 0) return fail(GF_ERR_ARG, "no pack in flight");
  DeviceGuard guard(s->ix->device);
  gf_stream::Slot& sl = s->slots[(size_t)s->tail];
  if (sl.zc_call) {
    const int rc = zc_wait(sl.zc, sl.st);
    if (rc != GF_OK) return rc;
    *out_n = zc_hits(sl.zc, sl.n, sl.read_id_base, out_hits, cap);
    sl.inflight = false;
    s->tail = (s->tail + 1) % s->depth;
    s->live -= 1;
    return GF_OK;
  }
  GF_HIP(hipEventSynchronize(sl.done));
  const int64_t total = sl.n > 0 ? *sl.h_total : 0;
  *out_n = total;
  const int64_t want = std::min(total, cap);
  const int64_t from_pin = std::min(want, std::min(sl.n, s->pin_cap));
  if (from_pin > 0) memcpy(out_hits, sl.h_hits, (size_t)from_pin * sizeof(gf_hit));
  if (want > from_pin)  // more hits than the pinned block holds: the rest straight from the device list
    GF_HIP(hipMemcpy(out_hits + from_pin, sl.d_hits + from_pin, (size_t)(want - from_pin) * sizeof(gf_hit),
                     hipMemcpyDeviceToHost));
  sl.inflight = false;
  s->tail = (s->tail + 1) % s->depth;
  s->live -= 1;
  return GF_OK;
}

int64_t gf_index_export(const gf_index* idx, int32_t what, void* out, int64_t cap) {
  if (!idx || cap < 0) return fail(GF_ERR_ARG, "null argument");
  const void* src = nullptr;
  int64_t bytes = 0;
  if (what == GF_EXPORT_GDU) {
    src = idx->d_gdu;
    bytes = 8 * (int64_t)idx->table.gd_words;
  } else if (what == GF_EXPORT_FILTER) {
    src = idx->d_bloom;
    bytes = 4 * (int64_t)idx->table.bloom_words;
  } else if (what == GF_EXPORT_LIN_BASE) {
    src = idx->d_lin_base;
    bytes = 4 * (int64_t)idx->table.n_genes;
  } else {
    return fail(GF_ERR_ARG, "gf_index_export: unknown array");
  }
  if (out && bytes > 0 && cap > 0) {
    DeviceGuard guard(idx->device);
    GF_HIP(hipDeviceSynchronize());
    GF_HIP(hipMemcpy(out, src, (size_t)std::min<int64_t>(cap, bytes), hipMemcpyDeviceToHost));
  }
  return bytes;
}

// ---- the exchange: per-rank hit lists -> the global list, one all-gather over RCCL ----
struct gf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

#define GF_NCCL(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) {                                                                       \
      char buf_[512];                                                                              \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return fail(GF_ERR_COMM, buf_);                                                              \
    }                                                                                              \
  } while (0)

static_assert(GF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "gf_comm id size");

int gf_comm_unique_id(void* out_id) {
  if (!out_id) return fail(GF_ERR_ARG, "out_id is null");
  ncclUniqueId id;
  GF_NCCL(ncclGetUniqueId(&id));
  memcpy(out_id, &id, sizeof id);
  return GF_OK;
}

int gf_comm_init(const void* id, int32_t rank, int32_t world, int32_t device, gf_comm** out) {
  if (!out) return fail(GF_ERR_ARG, "out is null");
  *out = nullptr;
  if (!id || world < 1 || world > GF_EXCH_MAX_WORLD || rank < 0 || rank >= world)
    return fail(GF_ERR_ARG, "bad id, rank or world (1..64 ranks)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(GF_ERR_NO_DEVICE, "no HIP device available (gfmatch has no CPU fallback)");
  if (device < 0) GF_HIP(hipGetDevice(&device));
  if (device >= ndev) return fail(GF_ERR_NO_DEVICE, "device ordinal out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(GF_ERR_NO_DEVICE, "hipSetDevice failed");
  std::unique_ptr<gf_comm> c(new gf_comm());
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  GF_NCCL(ncclCommInitRank(&c->comm, world, uid, rank));
  *out = c.release();
  return GF_OK;
}

int gf_comm_rank(const gf_comm* comm, int32_t* rank, int32_t* world) {
  if (!comm) return fail(GF_ERR_ARG, "null communicator");
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  return GF_OK;
}

void gf_comm_free(gf_comm* comm) {
  if (!comm) return;
  DeviceGuard guard(comm->device);
  if (comm->comm) (void)ncclCommDestroy(comm->comm);
  delete comm;
}

// send block | receive blocks, (cap + 1) records each
int64_t gf_allgather_workspace_bytes(int32_t world, int64_t cap) {
  if (world < 1 || cap < 0) return 0;
  return ((int64_t)world + 1) * (cap + 1) * (int64_t)sizeof(gf_hit) + 256;
}

static int exch_grid(int64_t records) { return (int)std::max<int64_t>(1, std::min<int64_t>((3 * records + 255) / 256, 1024)); }

int gf_pack_gathered_hits_device(const void* d_recv, int32_t world, int64_t cap, void* d_merged, void* d_totals, void* stream) {
  if (world < 1 || world > GF_EXCH_MAX_WORLD || cap < 0) return fail(GF_ERR_ARG, "bad world (1..64) or cap");
  if (!d_recv || !d_totals || (cap > 0 && !d_merged)) return fail(GF_ERR_ARG, "null device pointer");
  // standalone callers have no index or communicator to name the device: it is the one the receive buffer lives on
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_recv) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GF_ERR_ARG, "d_recv is not a device pointer");
  }
  DeviceGuard guard(attr.device);
  hipLaunchKernelGGL(gf_k_exch_pack, dim3(exch_grid((int64_t)world * cap)), dim3(256), 0, (hipStream_t)stream,
                     (const gf_hit*)d_recv, world, cap, (gf_hit*)d_merged, (int64_t*)d_totals);
  GF_HIP(hipGetLastError());
  return GF_OK;
}

int gf_allgather_hits_device(gf_comm* comm, const void* d_hits, const void* d_n_hits, int64_t cap, void* d_merged,
                             void* d_totals, void* d_workspace, void* stream) {
  if (!comm || cap < 0) return fail(GF_ERR_ARG, "null communicator or negative cap");
  if (!d_n_hits || !d_totals || !d_workspace || (cap > 0 && (!d_hits || !d_merged))) return fail(GF_ERR_ARG, "null device pointer");
  DeviceGuard guard(comm->device);
  hipStream_t st = (hipStream_t)stream;
  gf_hit* send = (gf_hit*)(((uintptr_t)d_workspace + 255) & ~(uintptr_t)255);
  gf_hit* recv = send + (cap + 1);
  hipLaunchKernelGGL(gf_k_exch_stage, dim3(exch_grid(cap)), dim3(256), 0, st, (const gf_hit*)d_hits, (const int64_t*)d_n_hits, cap, send);
  GF_HIP(hipGetLastError());
  GF_NCCL(ncclAllGather(send, recv, (size_t)(cap + 1) * sizeof(gf_hit), ncclChar, comm->comm, st));
  return gf_pack_gathered_hits_device(recv, comm->world, cap, d_merged, d_totals, stream);
}

int gf_index_trim(gf_index* idx) {
  if (!idx) return fail(GF_ERR_ARG, "null index");
  DeviceGuard guard(idx->device);
  {
    std::lock_guard<std::mutex> lk(idx->lane_mu);
    for (HostLane* L : idx->lanes) {
      if (L->busy) continue;  // in use by another thread: left alone
      if (L->arena) { GF_HIP(hipFree(L->arena)); L->arena = nullptr; L->arena_bytes = 0; }
      L->zc.release();
    }
  }
  // Calls queued on the workspaces' streams may still use them: wait for the whole device, not stream by
  // stream — a stream the caller has destroyed since is still a key here, and must not be touched.
  GF_HIP(hipDeviceSynchronize());
  for (int k = 1; k >= 0; --k) {
    WsPool& P = ws_pool(k == 1);
    std::vector<std::shared_ptr<Workspace>> mine;
    {
      std::lock_guard<std::mutex> lk(P.mu);
      for (auto& kv : P.ws)
        if (kv.first.device == idx->device && kv.second) mine.push_back(kv.second);
    }
    for (auto& e : mine) {
      std::lock_guard<std::mutex> lk(e->mu);
      if (e->base) {
        GF_HIP(hipFree(e->base));
        e->base = nullptr;
        e->bytes = 0;
      }
    }
  }
  block_trim(idx->device);  // device blocks of freed indexes
  return GF_OK;
}

int gf_set_map_variant(gf_index* idx, int32_t variant) {
  if (!idx || variant < 0 || variant > 2) return fail(GF_ERR_ARG, "bad variant");
  idx->map_variant = variant;
  return GF_OK;
}

int gf_set_pack_call_reads(gf_index* idx, int64_t reads) {
  if (!idx) return fail(GF_ERR_ARG, "null index");
  idx->pack_call_reads = reads < 0 ? -1 : reads;
  return GF_OK;
}

int gf_set_profiling(gf_index* idx, int32_t on) {
  if (!idx) return fail(GF_ERR_ARG, "null index");
  idx->profiling = on != 0;
  return GF_OK;
}

int gf_last_stage_ms(gf_index* idx, float out[4]) {
  if (!idx || !out) return fail(GF_ERR_ARG, "null argument");
  if (!idx->have_events || !idx->stages_recorded) return fail(GF_ERR_ARG, "no profiled flat-pipeline call yet");
  DeviceGuard guard(idx->device);
  std::lock_guard<std::mutex> lk(idx->prof_mu);
  GF_HIP(hipEventSynchronize(idx->ev_stage[4]));
  for (int k = 0; k < 4; ++k) GF_HIP(hipEventElapsedTime(&out[k], idx->ev_stage[k], idx->ev_stage[k + 1]));
  return GF_OK;
}

float gf_last_map_kernel_ms(gf_index* idx) {
  if (!idx || !idx->have_events || !idx->recorded) return -1.0f;
  DeviceGuard guard(idx->device);
  std::lock_guard<std::mutex> lk(idx->prof_mu);
  if (hipEventSynchronize(idx->ev1) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, idx->ev0, idx->ev1) != hipSuccess) return -1.0f;
  return ms;
}

}  // extern "C"
