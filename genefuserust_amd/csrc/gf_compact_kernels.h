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

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_compact_count(const uint8_t* __restrict__ counts,
                                                                  int64_t n,
                                                                  uint32_t* __restrict__ tile_counts) {
  __shared__ int s_wave[4];
  const int64_t r0 = (int64_t)blockIdx.x * GF_CTILE + (int64_t)threadIdx.x * GF_CPER;
  int c = 0;
  for (int k = 0; k < GF_CPER; ++k)
    if (r0 + k < n) c += gf_is_hit(counts[r0 + k]);
  int total;
  gf_block_exclusive_scan(c, s_wave, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = (uint32_t)total;
}

// exclusive scan of the tile totals by ONE block: 32 consecutive totals per thread (eight 16-byte loads in
// flight, then serial), wave scans by shuffle, the 16 wave sums through LDS — 32768 totals per round, two
// barriers.  (Eight totals per thread made five dependent rounds of the pair kernels' 39 K tiles: 62 us per scan,
// ten scans per pack.)  Block b of the launch scans array b: the callers' scans come in pairs (bytes and reads).
struct GfScanJob {
  const uint32_t* tile_counts;
  int64_t* tile_offsets;
  int64_t* d_total;
};
struct GfScanJobs { GfScanJob j[2]; };

__global__ __launch_bounds__(1024) void gf_k_compact_scan(GfScanJobs jobs, int64_t ntiles) {
  constexpr int PER = 32;
  const uint32_t* __restrict__ tile_counts = jobs.j[blockIdx.x].tile_counts;
  int64_t* __restrict__ tile_offsets = jobs.j[blockIdx.x].tile_offsets;
  int64_t* __restrict__ d_total = jobs.j[blockIdx.x].d_total;
  __shared__ long long s_wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec_ok = (((uintptr_t)tile_counts) & 15u) == 0;
  long long run = 0;  // sum of all earlier rounds (the same in every thread)
  for (int64_t b0 = 0; b0 < ntiles; b0 += 1024 * PER) {
    const int64_t t0 = b0 + (int64_t)threadIdx.x * PER;
    uint32_t v[PER];
    if (vec_ok && t0 + PER <= ntiles) {
      const uint4* q = (const uint4*)(tile_counts + t0);
#pragma unroll
      for (int k = 0; k < PER / 4; ++k) {
        const uint4 x = q[k];
        v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) v[k] = t0 + k < ntiles ? tile_counts[t0 + k] : 0u;
    }
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) mine += v[k];
    long long x = mine;  // inclusive scan of the threads' sums within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    long long base = run, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const long long ws = s_wsum[w];
      if (w < wave) base += ws;
      all += ws;
    }
    long long pos = base + x - mine;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (t0 + k < ntiles) tile_offsets[t0 + k] = pos;
      pos += v[k];
    }
    run += all;
    __syncthreads();  // s_wsum is rewritten in the next round
  }
  if (threadIdx.x == 0) *d_total = run;
}

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_compact_write(
    const uint8_t* __restrict__ counts, const gf_seqmatch* __restrict__ matches, int64_t n,
    int64_t read_id_base, const int64_t* __restrict__ tile_offsets, gf_hit* __restrict__ hits,
    int64_t cap) {
  __shared__ int s_wave[4];
  const int64_t r0 = (int64_t)blockIdx.x * GF_CTILE + (int64_t)threadIdx.x * GF_CPER;
  int c = 0;
  for (int k = 0; k < GF_CPER; ++k)
    if (r0 + k < n) c += gf_is_hit(counts[r0 + k]);
  int total;
  int64_t pos = tile_offsets[blockIdx.x] + gf_block_exclusive_scan(c, s_wave, &total);
  if (!c) return;
  for (int k = 0; k < GF_CPER; ++k) {
    const int64_t r = r0 + k;
    if (r >= n) break;
    const uint8_t cn = counts[r];
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
