// The tail of FusionMapper::map_read on the device (VERDICT r02 "missing 4"): make_match
// (fusion_mapper.rs:154-194) and calc_distance / calc_ed (:196-251) with edit_distance (edit_distance.rs:12-197) for
// the records of gf_scan_pairs_device while they and their reads are still in HBM — the ~0.06 % of reads that come
// back with two segments in the required direction.
//
// One wavefront per (record, side of the break).  Levenshtein distance by Hyyro's block-based bit-vector recurrence —
// the algorithm of edit_distance.rs:12-92 and of the host form in gfmatch.hip (ed_bitparallel), for any number of
// 64-symbol blocks: the distance is a number, any exact algorithm gives it.  The wavefront is what makes it cheap:
// lane j holds symbols j, 64 + j, 128 + j .. of the pattern, so the match mask of a text symbol against block r is ONE
// ballot; the block updates are the same in every lane (scalar registers and a few LDS words per block).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"
#include "gf_map_kernels.h"

#define GF_TAIL_MAX_BLOCKS 64  // GF_MAX_READ_LEN / 64

__device__ __forceinline__ uint32_t gf_tail_complement(uint32_t c) {  // sequence.rs:51-59
  switch (c) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

// pattern = `m` symbols read through pat(i) (any byte; 256 = none), text = t[0 .. n); s_v = 4 * GF_TAIL_MAX_BLOCKS words of LDS
template <typename PatFn>
__device__ int64_t gf_wave_edit_distance(PatFn pat, int m, const uint8_t* __restrict__ t, int n, uint64_t* s_v, int lane) {
  if (m == 0) return n;
  if (n == 0) return m;
  const int tmax = (m - 1) >> 6, tlen = m - 64 * tmax;
  uint64_t* vp = s_v;
  uint64_t* vn = s_v + GF_TAIL_MAX_BLOCKS;
  uint64_t* hp = s_v + 2 * GF_TAIL_MAX_BLOCKS;
  uint64_t* hn = s_v + 3 * GF_TAIL_MAX_BLOCKS;
  for (int r = lane; r <= tmax; r += 64) {
    vp[r] = r == tmax ? (tlen == 64 ? ~0ull : ((1ull << tlen) - 1)) : ~0ull;
    vn[r] = 0;
  }
  gf_wave_lds_sync();  // the wave's own LDS writes
  const uint64_t top = 1ull << (tlen - 1), msb = 1ull << 63;
  int64_t d = m;
  for (int i = 0; i < n; ++i) {
    const uint32_t c = t[i];
    uint64_t hp_prev = 0, hn_prev = 0;
    for (int r = 0; r <= tmax; ++r) {
      const int pi = 64 * r + lane;
      uint64_t x = __ballot(pi < m && pat(pi) == c);
      const uint64_t vpr = vp[r], vnr = vn[r];
      const bool carry_n = r > 0 && (hn_prev & msb);
      if (carry_n) x |= 1ull;
      const uint64_t d0 = (((x & vpr) + vpr) ^ vpr) | x | vnr;
      const uint64_t hpr = vnr | ~(d0 | vpr);
      const uint64_t hnr = d0 & vpr;
      uint64_t y = hpr << 1;
      if (r == 0 || (hp_prev & msb)) y |= 1ull;
      uint64_t nvp = (hnr << 1) | ~(d0 | y);
      if (carry_n) nvp |= 1ull;
      if (lane == 0) { vp[r] = nvp; vn[r] = d0 & y; }
      hp_prev = hpr; hn_prev = hnr;
      if (r == tmax) {
        if (hpr & top) d += 1;
        else if (hnr & top) d -= 1;
      }
    }
    gf_wave_lds_sync();
  }
  (void)hp; (void)hn;
  return d;
}

// fusion_mapper.rs:225-251 for one side: `seq` = the read's part (len symbols), [start, end] on the gene
__device__ int32_t gf_wave_calc_ed(const uint8_t* __restrict__ fs, int64_t fs_len, const uint8_t* __restrict__ seq, int32_t len,
                                   int32_t start, int32_t end, uint64_t* s_v, int lane) {
  if ((start >= 0 && end <= 0) || (start <= 0 && end >= 0)) return -1;  // not on one strand
  const int64_t as = start < 0 ? -(int64_t)start : start, ae = end < 0 ? -(int64_t)end : end;
  if (as >= fs_len || ae >= fs_len) return -2;
  if (start < 0) {  // the read's reverse complement against the forward strand
    const int32_t s2 = -end, e2 = -start;
    auto pat = [&](int i) -> uint32_t { return gf_tail_complement(seq[len - 1 - i]); };
    return (int32_t)gf_wave_edit_distance(pat, len, fs + s2, e2 - s2 + 1, s_v, lane);
  }
  auto pat = [&](int i) -> uint32_t { return seq[i]; };
  return (int32_t)gf_wave_edit_distance(pat, len, fs + start, end - start + 1, s_v, lane);
}

// status[k]: GF_RM_MATCH, or GF_ERR_ARG when a record names a contig outside the gene list or segments outside its read
__global__ __launch_bounds__(256) void gf_k_pair_hits_finish(const gf_pair_hit* __restrict__ hits, const int64_t* __restrict__ d_n,
                                                             int64_t cap, const uint8_t* __restrict__ hit_bases,
                                                             const uint8_t* __restrict__ cat, const uint32_t* __restrict__ gene_off,
                                                             const uint32_t* __restrict__ gene_len, int32_t n_genes,
                                                             gf_readmatch* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ uint64_t s_v_all[4][4 * GF_TAIL_MAX_BLOCKS];
  const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint64_t* s_v = s_v_all[wib];
  int64_t n = *d_n;
  if (n > cap) n = cap;
  for (int64_t job = (int64_t)blockIdx.x * 4 + wib; job < 2 * n; job += (int64_t)gridDim.x * 4) {
    const int64_t k = job >> 1;
    const int side = (int)(job & 1);
    const gf_pair_hit h = hits[k];
    gf_seqmatch left = h.m[0], right = h.m[1];
    if (left.seq_start > right.seq_start) { const gf_seqmatch t = left; left = right; right = t; }
    const int32_t len = h.read_len;
    const int32_t read_break = (left.seq_end + right.seq_start) / 2;  // fusion_mapper.rs:173
    const int32_t left_len = read_break + 1, right_len = len - (read_break + 1);
    const bool bad = left.contig < 0 || left.contig >= n_genes || right.contig < 0 || right.contig >= n_genes ||
                     left_len < 0 || right_len < 0 || left_len > len;
    if (bad) {
      if (lane == 0 && side == 0) status[k] = GF_ERR_ARG;
      continue;
    }
    left.position += read_break;       // :177-178
    right.position += read_break + 1;
    const uint8_t* seq = hit_bases + h.seq_offset;
    int32_t dist;
    if (side == 0)
      dist = gf_wave_calc_ed(cat + gene_off[left.contig], (int64_t)gene_len[left.contig], seq, left_len,
                             left.position - left_len + 1, left.position, s_v, lane);
    else
      dist = gf_wave_calc_ed(cat + gene_off[right.contig], (int64_t)gene_len[right.contig], seq + read_break + 1, right_len,
                             right.position, right.position + right_len - 1, s_v, lane);
    if (lane == 0) {
      gf_readmatch* o = out + k;
      if (side == 0) {
        o->read_break = read_break;
        o->gap = right.seq_start - left.seq_end - 1;  // :180
        o->left_contig = left.contig;
        o->left_position = left.position;
        o->right_contig = right.contig;
        o->right_position = right.position;
        o->left_distance = dist;
        status[k] = GF_RM_MATCH;
      } else {
        o->right_distance = dist;
      }
    }
  }
}
