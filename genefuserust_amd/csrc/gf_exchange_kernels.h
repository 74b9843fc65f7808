// The path's one exchange: the per-rank ordered hit lists merged into the global ordered list on every rank
// (SURVEY.md §8e, C2).  The reference has no analogue — its consumer threads push ReadMatches under one mutex
// (fusion_mapper.rs:253-275) and the name tiebreak repairs the order afterwards (read_match.rs:227); here shards
// are contiguous read ranges and every list is ascending, so rank order IS read order.
//
// One fixed-capacity all-gather (ncclAllGather of (cap + 1) records per rank; record 0 of a block carries the
// rank's count) so that nothing goes through the host between the mapping and the merged list:
//   gf_k_exch_stage   count + the first min(count, cap) records -> the rank's send block
//   (ncclAllGather)
//   gf_k_exch_pack    valid prefixes of the world blocks -> one list in rank order, total, overflow flag
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"

#define GF_EXCH_MAX_WORLD 64

// a gf_hit as three 16-byte vectors
typedef uint32_t gf_exch_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gf_k_exch_stage(const gf_hit* __restrict__ hits, const int64_t* __restrict__ n_hits,
                                                       int64_t cap, gf_hit* __restrict__ send) {
  const int64_t n = *n_hits;
  const int64_t m = n < cap ? (n < 0 ? 0 : n) : cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gf_hit h;
    h.read_id = n;  // the header record: this rank's count (may exceed cap: reported as overflow by the pack)
    h.n = 0; h.pad = 0;
    h.m[0].seq_start = h.m[0].seq_end = h.m[0].position = 0; h.m[0].contig = 0; h.m[0].pad = 0;
    h.m[1] = h.m[0];
    send[0] = h;
  }
  const gf_exch_u32x4* src = (const gf_exch_u32x4*)hits;
  gf_exch_u32x4* dst = (gf_exch_u32x4*)(send + 1);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < 3 * m; v += (int64_t)gridDim.x * blockDim.x) dst[v] = src[v];
}

// recv = world blocks of (cap + 1) records.  merged must hold world * cap records.  totals[0] = records in the
// merged list, totals[1] = 1 when some rank had more than cap (its list is cut at cap: run that batch again with
// a larger capacity), totals[2 + r] = rank r's own count.
__global__ __launch_bounds__(256) void gf_k_exch_pack(const gf_hit* __restrict__ recv, int32_t world, int64_t cap,
                                                      gf_hit* __restrict__ merged, int64_t* __restrict__ totals) {
  __shared__ int64_t s_start[GF_EXCH_MAX_WORLD + 1];
  if (threadIdx.x == 0) {
    int64_t run = 0, over = 0;
    for (int r = 0; r < world; ++r) {
      int64_t c = recv[(int64_t)r * (cap + 1)].read_id;
      if (blockIdx.x == 0) totals[2 + r] = c;
      if (c > cap) { over = 1; c = cap; }
      if (c < 0) c = 0;
      s_start[r] = run;
      run += c;
    }
    s_start[world] = run;
    if (blockIdx.x == 0) { totals[0] = run; totals[1] = over; }
  }
  __syncthreads();
  const gf_exch_u32x4* src = (const gf_exch_u32x4*)recv;
  gf_exch_u32x4* dst = (gf_exch_u32x4*)merged;
  for (int r = 0; r < world; ++r) {
    const int64_t cnt = s_start[r + 1] - s_start[r];
    const int64_t s0 = 3 * ((int64_t)r * (cap + 1) + 1), d0 = 3 * s_start[r];
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < 3 * cnt; v += (int64_t)gridDim.x * blockDim.x) dst[d0 + v] = src[s0 + v];
  }
}
