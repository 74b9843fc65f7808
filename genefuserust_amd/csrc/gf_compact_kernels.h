// K4 — ordered compaction of the dense per-read result into gf_hit records.
//
// The reference returns a Vec<SeqMatch> per read (indexer.rs:252) and the caller
// keeps only reads with two segments (fusion_mapper.rs:107-115); >99 % of reads
// return nothing.  The mapping kernel writes a dense count byte per read; these
// three small kernels turn the non-empty entries into a list in ascending read
// order (deterministic: two-level exclusive scan, no atomics), which is what
// crosses PCIe / xGMI.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"

#define GF_CTILE 4096
#define GF_CTHREADS 256
#define GF_CPER (GF_CTILE / GF_CTHREADS)

__device__ __forceinline__ int gf_is_hit(uint8_t c) { return c >= 1 && c <= 2; }

__device__ __forceinline__ int gf_block_exclusive_scan(int v, int* s_wave, int* total) {
  // 256 threads = 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  return base + x - v;
}

// the thread's GF_CPER (= 16) counts: one 16-byte load where the address allows it (sixteen byte loads took most of
// the two sweeps' 24 + 30 us per 20 M reads), bytes beyond n read as 0
__device__ __forceinline__ void gf_counts16(const uint8_t* __restrict__ counts, int64_t r0, int64_t n, uint8_t (&c)[GF_CPER]) {
  static_assert(GF_CPER == 16, "one uint4 per thread");
  if (r0 + GF_CPER <= n && (((uintptr_t)(counts + r0)) & 15u) == 0) {
    const uint4 q = *(const uint4*)(counts + r0);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < GF_CPER; ++k) c[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
  } else {
#pragma unroll
    for (int k = 0; k < GF_CPER; ++k) c[k] = r0 + k < n ? counts[r0 + k] : (uint8_t)0;
  }
}

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_compact_count(const uint8_t* __restrict__ counts,
                                                                  int64_t n,
                                                                  uint32_t* __restrict__ tile_counts) {
  __shared__ int s_wave[4];
  const int64_t r0 = (int64_t)blockIdx.x * GF_CTILE + (int64_t)threadIdx.x * GF_CPER;
  uint8_t cv[GF_CPER];
  gf_counts16(counts, r0, n, cv);
  int c = 0;
#pragma unroll
  for (int k = 0; k < GF_CPER; ++k) c += gf_is_hit(cv[k]);
  int total;
  gf_block_exclusive_scan(c, s_wave, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = (uint32_t)total;
}

// exclusive scan of the tile totals by ONE block, 16384 totals per round: thread t takes totals t, t + 1024, ..
// (sixteen rows of 1024, every load and store of a wavefront one contiguous run), each row is scanned inside
// its wavefronts by shuffles, the 16 x 16 wavefront sums go through LDS where the first wavefront turns them into
// their exclusive prefix, and every total's offset is its row-and-wavefront base plus its place in the
// wavefront.  Three barriers per round.  (A thread owning consecutive totals — eight, then thirty-two — stored
// its offsets 64 lanes to 64 different lines: 58-62 us for the pair kernels' 39 K tiles, most of it the one
// CU's store path; ten such scans per pack.)  Block b of the launch scans array b: the callers' scans come in
// pairs (bytes and reads).
struct GfScanJob {
  const uint32_t* tile_counts;
  int64_t* tile_offsets;
  int64_t* d_total;
};
struct GfScanJobs { GfScanJob j[2]; };

#define GF_SCAN_ROWS 16
#define GF_SCAN_ROUND (1024 * GF_SCAN_ROWS)  // totals per round of a block

// one round: totals b0 .. b0 + GF_SCAN_ROUND - 1 of a 1024-thread block; offsets = run + exclusive prefix; returns
// the round's sum (the same in every thread).  s_w: GF_SCAN_ROWS * 16 + 1 long longs of LDS.
__device__ __forceinline__ long long gf_scan_round(const uint32_t* __restrict__ tile_counts, int64_t ntiles, int64_t b0,
                                                   long long run, int64_t* __restrict__ tile_offsets, long long* s_w) {
  constexpr int ROWS = GF_SCAN_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t v[ROWS];
  long long x[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int64_t e = b0 + (int64_t)k * 1024 + threadIdx.x;
    v[k] = e < ntiles ? tile_counts[e] : 0u;
  }
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    long long y = v[k];  // inclusive scan inside the wavefront
    if (b0 + (int64_t)k * 1024 < ntiles) {  // (the same for the whole block: rows past the end hold zeros)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long long z = __shfl_up(y, o);
        if (lane >= o) y += z;
      }
    }
    x[k] = y;
    if (lane == 63) s_w[k * 16 + wave] = y;
  }
  __syncthreads();
  if (wave == 0) {  // 256 sums, four per lane, in (row, wavefront) order
    long long a[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = s_w[4 * lane + i];
      mine += a[i];
    }
    long long y = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long z = __shfl_up(y, o);
      if (lane >= o) y += z;
    }
    long long pos = y - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_w[4 * lane + i] = pos;
      pos += a[i];
    }
    if (lane == 63) s_w[ROWS * 16] = y;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int64_t e = b0 + (int64_t)k * 1024 + threadIdx.x;
    if (e < ntiles) tile_offsets[e] = run + s_w[k * 16 + wave] + x[k] - (long long)v[k];
  }
  const long long total = s_w[ROWS * 16];
  __syncthreads();  // s_w is rewritten in the next round
  return total;
}

__global__ __launch_bounds__(1024) void gf_k_compact_scan(GfScanJobs jobs, int64_t ntiles) {
  const uint32_t* __restrict__ tile_counts = jobs.j[blockIdx.x].tile_counts;
  int64_t* __restrict__ tile_offsets = jobs.j[blockIdx.x].tile_offsets;
  int64_t* __restrict__ d_total = jobs.j[blockIdx.x].d_total;
  __shared__ long long s_w[GF_SCAN_ROWS * 16 + 1];  // [row][wavefront]: sums, then their exclusive prefix; the total
  long long run = 0;  // sum of all earlier rounds (the same in every thread)
  for (int64_t b0 = 0; b0 < ntiles; b0 += GF_SCAN_ROUND) run += gf_scan_round(tile_counts, ntiles, b0, run, tile_offsets, s_w);
  if (threadIdx.x == 0) *d_total = run;
}

// Scans of hundreds of thousands of totals (the newline counts of a FASTQ text: one per 16 KB) in three launches
// instead of dozens of dependent rounds of one block: block b scans round b on its own and leaves the round's sum,
// gf_k_compact_scan scans those sums, gf_k_compact_scan_add adds a round's base to its offsets.
__global__ __launch_bounds__(1024) void gf_k_compact_scan_rounds(const uint32_t* __restrict__ tile_counts, int64_t ntiles,
                                                                 int64_t* __restrict__ tile_offsets,
                                                                 uint32_t* __restrict__ round_sums) {
  __shared__ long long s_w[GF_SCAN_ROWS * 16 + 1];
  const long long t = gf_scan_round(tile_counts, ntiles, (int64_t)blockIdx.x * GF_SCAN_ROUND, 0, tile_offsets, s_w);
  if (threadIdx.x == 0) round_sums[blockIdx.x] = (uint32_t)t;  // (the caller's totals keep a round's sum below 2^32)
}

__global__ __launch_bounds__(1024) void gf_k_compact_scan_add(int64_t* __restrict__ tile_offsets, int64_t ntiles,
                                                              const int64_t* __restrict__ round_offsets) {
  const long long base = round_offsets[blockIdx.x];
  if (base == 0) return;
#pragma unroll
  for (int k = 0; k < GF_SCAN_ROWS; ++k) {
    const int64_t e = (int64_t)blockIdx.x * GF_SCAN_ROUND + (int64_t)k * 1024 + threadIdx.x;
    if (e < ntiles) tile_offsets[e] += base;
  }
}

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_compact_write(
    const uint8_t* __restrict__ counts, const gf_seqmatch* __restrict__ matches, int64_t n,
    int64_t read_id_base, const int64_t* __restrict__ tile_offsets, gf_hit* __restrict__ hits,
    int64_t cap) {
  __shared__ int s_wave[4];
  const int64_t r0 = (int64_t)blockIdx.x * GF_CTILE + (int64_t)threadIdx.x * GF_CPER;
  uint8_t cv[GF_CPER];
  gf_counts16(counts, r0, n, cv);
  int c = 0;
#pragma unroll
  for (int k = 0; k < GF_CPER; ++k) c += gf_is_hit(cv[k]);
  int total;
  int64_t pos = tile_offsets[blockIdx.x] + gf_block_exclusive_scan(c, s_wave, &total);
  if (!c) return;
#pragma unroll
  for (int k = 0; k < GF_CPER; ++k) {
    const int64_t r = r0 + k;
    const uint8_t cn = cv[k];  // (0 beyond n)
    if (!gf_is_hit(cn)) continue;
    if (pos < cap) {
      gf_hit h;
      h.read_id = read_id_base + r;
      h.n = cn;
      h.pad = 0;
      h.m[0] = matches[2 * r];
      if (cn == 2) {
        h.m[1] = matches[2 * r + 1];
      } else {
        h.m[1].seq_start = 0; h.m[1].seq_end = 0; h.m[1].position = 0; h.m[1].contig = 0; h.m[1].pad = 0;
      }
      hits[pos] = h;
    }
    ++pos;
  }
}
