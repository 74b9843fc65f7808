// SURVEY.md §8(f)-2 — SequenceReadPair::fast_merge (src/core/read.rs:313-440) on the
// device: the step before the hot path.  Pairs are independent.
//
// rc_right = reverse complement of R2 (anything outside ACGTacgt -> 'N', output upper case,
// sequence.rs:22-60), its quality reversed.  The smallest overlap olen >= 30 is taken
// for which every mismatch between R1's tail and rc_right's head is a "low quality"
// mismatch (one base >= Q30 i.e. >= '?', the other <= Q15 i.e. <= '0') and there are
// fewer than three of them (read.rs:339-367; the loop's `diff > low_qual_diff ||
// low_qual_diff >= 3` is order-independent: no high-quality mismatch, at most two
// low-quality ones).  merged = R1[0, len1-olen) + rc_right, overlap corrected (:402-428).
//
// Two facts make the search cheap:
//   * three mismatching columns reject an overlap whatever their qualities (either one of
//     them is a high-quality mismatch, or there are three low-quality ones), so the search
//     runs on bases alone and touches qualities only for overlaps with <= 2 mismatches;
//   * in the 2-bit packed form a candidate overlap is tested 16 columns per XOR.
//
//   gf_k_pack       (gf_pipe_kernels.h) R1 bytes -> packed stream, coalesced
//   gf_k_pack_rc    R2 bytes -> packed stream of the whole buffer reversed and complemented:
//                   rc(R2 of pair p) is a contiguous piece of it, no per-pair reversal
//   gf_k_merge_find thread per pair: slide R1's packed words past the first 16 bases of
//                   rc(R2); candidates with <= 2 mismatches there get the full-overlap
//                   count (still packed) and, with <= 2 in total, the quality test
//   gf_k_merge_write wave per merged pair: assembles the merged read and its quality
// Pairs the packed form cannot decide exactly (a read longer than the kernel's word
// budget, outside the packed stream, or both reads holding a byte outside A/C/G/T where
// 'N' == 'N' could make two such bytes equal) take gf_merge_find_bytes, the plain byte
// loop.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf_pipe_kernels.h"

#define GF_MERGE_MIN_OVERLAP 30

__device__ __forceinline__ uint8_t gf_complement(uint8_t b) {
  switch (b) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

__device__ __forceinline__ bool gf_lowq_pair(uint8_t a, uint8_t b) {
  return (a >= '?' && b <= '0') || (a <= '0' && b >= '?');
}

// The byte loop of read.rs:339-367.  Returns olen (0 = no overlap qualifies), sets diff.
__device__ inline int gf_merge_find_bytes(const uint8_t* s1, const uint8_t* q1, int len1, const uint8_t* s2,
                                          const uint8_t* q2, int len2, int& diff) {
  const int lim = len1 < len2 ? len1 : len2;
  for (int olen = GF_MERGE_MIN_OVERLAP; olen <= lim; ++olen) {
    const int offset = len1 - olen;
    int d = 0;
    bool ok = true;
    for (int i = 0; i < olen; ++i) {
      if (s1[offset + i] != gf_complement(s2[len2 - 1 - i])) {
        d += 1;
        if (!gf_lowq_pair(q1[offset + i], q2[len2 - 1 - i]) || d >= 3) {
          ok = false;
          break;
        }
      }
    }
    if (ok) {
      diff = d;
      return olen;
    }
  }
  diff = 0;
  return 0;
}

// 4 ASCII bases -> 8 code bits and 4 "bad" bits, case-insensitive: valid = ACGTacgt, the
// bytes whose reverse complement is not 'N' (sequence.rs:22-60).
__device__ __forceinline__ void gf_convert4_bits_nocase(uint32_t x, uint32_t& code8, uint32_t& bad4) {
  uint32_t y = (x >> 1) & 0x03030303u;
  code8 = (y * 0x01041040u) >> 24;
  uint32_t bad = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t b = (x >> (8 * j)) & 0xFFu;
    uint32_t v = (((b & 0xC0u) == 0x40u) ? 1u : 0u) & (0x0010008Au >> (b & 31u));
    bad |= (v ^ 1u) << j;
  }
  bad4 = bad;
}

// chunks of 16 bytes the packed stream of a buffer covers (same formula in every kernel)
__device__ __forceinline__ uint64_t gf_stream_chunks(const uint8_t* bases, const int64_t* offsets, int64_t n,
                                                     uint64_t cap_chunks) {
  const uintptr_t a0 = gf_stream_origin(bases, offsets);
  const uintptr_t end = (uintptr_t)(bases + offsets[n]);
  uint64_t chunks = end > a0 ? (uint64_t)((end - a0 + 15) >> 4) : 0;
  return chunks > cap_chunks ? cap_chunks : chunks;
}

// ---- K_pack_rc: thread per 16 bytes of R2; word (chunks-1-t) of the output = the chunk's
// bases reversed and complemented (A0 C1 T2 G3: complement = code ^ 2) ----
__global__ __launch_bounds__(256) void gf_k_pack_rc(const uint8_t* __restrict__ bases,
                                                    const int64_t* __restrict__ offsets, int64_t n,
                                                    uint64_t cap_chunks, uint32_t* __restrict__ pkg,
                                                    uint16_t* __restrict__ ivg16) {
  const uintptr_t a0 = gf_stream_origin(bases, offsets);
  const uint64_t chunks = gf_stream_chunks(bases, offsets, n, cap_chunks);
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < chunks;
       t += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 q = *(const uint4*)(a0 + 16 * t);
    uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
    gf_convert4_bits_nocase(q.x, c0, b0);
    gf_convert4_bits_nocase(q.y, c1, b1);
    gf_convert4_bits_nocase(q.z, c2, b2);
    gf_convert4_bits_nocase(q.w, c3, b3);
    const uint32_t c = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    const uint32_t b = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
    uint32_t x = __brev(c);  // fields reversed, and the two bits of each field swapped
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    pkg[chunks - 1 - t] = x ^ 0xAAAAAAAAu;
    ivg16[chunks - 1 - t] = (uint16_t)(__brev(b) >> 16);
  }
}

// even-bit mask of the bases below L in word j of a read
__device__ __forceinline__ uint32_t gf_len_mask2(int L, int j) {
  const int k = L - 16 * j;
  return k >= 16 ? 0x55555555u : (k <= 0 ? 0u : (((1u << (2 * k)) - 1u) & 0x55555555u));
}

// ---- K_merge_find: thread per pair ----
template <int PW>
__global__ __launch_bounds__(256) void gf_k_merge_find(GfStream S1, GfStream S2, const uint8_t* __restrict__ l_bases,
                                                       const uint8_t* __restrict__ l_quals,
                                                       const int64_t* __restrict__ l_off,
                                                       const uint8_t* __restrict__ r_bases,
                                                       const uint8_t* __restrict__ r_quals,
                                                       const int64_t* __restrict__ r_off, int64_t n,
                                                       int32_t* __restrict__ out_len, int32_t* __restrict__ out_diff) {
  // R1's words, indexed dynamically by the full-overlap count ([word][thread]: own column only)
  __shared__ uint32_t s_a[(PW + 1) * 256];
  __shared__ uint32_t s_ia[(PW + 1) * 256];
  const uintptr_t a1 = gf_stream_origin(l_bases, l_off);
  const uintptr_t a2 = gf_stream_origin(r_bases, r_off);
  const uint64_t chunks2 = gf_stream_chunks(r_bases, r_off, n, S2.cap_bases >> 4);
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo1 = l_off[p], lo2 = r_off[p];
    const int64_t len1_64 = l_off[p + 1] - lo1, len2_64 = r_off[p + 1] - lo2;
    const int64_t lim64 = len1_64 < len2_64 ? len1_64 : len2_64;
    if (lim64 < GF_MERGE_MIN_OVERLAP) {
      out_len[p] = 0;
      out_diff[p] = 0;
      continue;
    }
    const uint8_t* s1 = l_bases + lo1;
    const uint8_t* q1 = l_quals + lo1;
    const uint8_t* s2 = r_bases + lo2;
    const uint8_t* q2 = r_quals + lo2;
    const uint64_t pos1 = (uint64_t)((uintptr_t)s1 - a1);
    const uint64_t x0 = (uint64_t)((uintptr_t)s2 - a2);  // R2's first byte in its buffer's stream
    bool bytes_path = len1_64 > 16 * PW || len2_64 > 16 * PW || pos1 + (uint64_t)len1_64 + 64 > S1.cap_bases ||
                      x0 + (uint64_t)len2_64 > 16 * chunks2;
    int found = 0, diff = 0;
    const int len1 = (int)len1_64, len2 = (int)len2_64;
    if (!bytes_path) {
      const uint64_t pos2 = 16 * chunks2 - (x0 + (uint64_t)len2);  // rc(R2) in the reversed stream
      uint32_t A[PW], IA[PW], B[PW], IB[PW];
      gf_load_read_words<PW>(S1, pos1, len1, A, IA);
      gf_load_read_words<PW>(S2, pos2, len2, B, IB);
      uint32_t anyA = 0, anyB = 0;
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        anyA |= IA[j] & gf_len_mask2(len1, j);
        anyB |= IB[j] & gf_len_mask2(len2, j);
        s_a[j * 256 + threadIdx.x] = A[j];
        s_ia[j * 256 + threadIdx.x] = IA[j];
      }
      s_a[PW * 256 + threadIdx.x] = 0;
      s_ia[PW * 256 + threadIdx.x] = 0x55555555u;
      // a byte outside A/C/G/T in R1 and one outside ACGTacgt in R2 could be equal ('N' vs the
      // 'N' of the reverse complement): only the byte loop knows
      bytes_path = anyA && anyB;
      if (!bytes_path) {
        const int lim = len1 < len2 ? len1 : len2;
        const int o_hi = len1 - GF_MERGE_MIN_OVERLAP, o_lo = len1 - lim;  // offsets to try, high to low
        const uint32_t b0 = B[0], ib0 = IB[0];
        bool done = false;
#pragma unroll
        for (int j = PW - 1; j >= 0; --j) {
          const uint32_t alo = A[j], ahi = j + 1 < PW ? A[j + 1] : 0u;
          const uint32_t ilo = IA[j], ihi = j + 1 < PW ? IA[j + 1] : 0x55555555u;
          if (16 * j <= o_hi && 16 * j + 15 >= o_lo) {
            for (int s = 15; s >= 0; --s) {
              const int o = 16 * j + s;
              if (done || o > o_hi || o < o_lo) continue;
              // first 16 columns: R1 bases o..o+15 against rc(R2) bases 0..15
              const uint32_t w = __builtin_amdgcn_alignbit(ahi, alo, 2u * (uint32_t)s);
              const uint32_t iw = __builtin_amdgcn_alignbit(ihi, ilo, 2u * (uint32_t)s);
              const uint32_t x = w ^ b0;
              const uint32_t m = ((x | (x >> 1)) | iw | ib0) & 0x55555555u;
              if (__popc(m) > 2) continue;
              // all columns of this overlap, 16 per word
              const int olen = len1 - o;
              int cnt = 0, c0 = -1, c1 = -1;
#pragma unroll
              for (int jj = 0; jj < PW; ++jj) {
                const uint32_t cm = gf_len_mask2(olen, jj);
                if (cm) {
                  const int pos = o + 16 * jj;
                  const int wi = pos >> 4;
                  const uint32_t sh = 2u * (uint32_t)(pos & 15);
                  const uint32_t ww = __builtin_amdgcn_alignbit(s_a[(wi + 1) * 256 + threadIdx.x],
                                                                s_a[wi * 256 + threadIdx.x], sh);
                  const uint32_t iww = __builtin_amdgcn_alignbit(s_ia[(wi + 1) * 256 + threadIdx.x],
                                                                 s_ia[wi * 256 + threadIdx.x], sh);
                  const uint32_t xx = ww ^ B[jj];
                  uint32_t mm = ((xx | (xx >> 1)) | iww | IB[jj]) & cm;
                  cnt += __popc(mm);
                  while (mm && c1 < 0) {
                    const int col = 16 * jj + (__builtin_ctz(mm) >> 1);
                    if (c0 < 0) c0 = col; else c1 = col;
                    mm &= mm - 1;
                  }
                }
              }
              if (cnt > 2) continue;
              bool ok = true;
              if (c0 >= 0) ok = gf_lowq_pair(q1[o + c0], q2[len2 - 1 - c0]);
              if (ok && c1 >= 0) ok = gf_lowq_pair(q1[o + c1], q2[len2 - 1 - c1]);
              if (ok) {
                found = olen;
                diff = cnt;
                done = true;
              }
            }
          }
        }
      }
    }
    if (bytes_path) found = gf_merge_find_bytes(s1, q1, len1, s2, q2, len2, diff);
    out_len[p] = found ? len1 - found + len2 : 0;
    out_diff[p] = diff;
  }
}

// Every pair through the byte loop (reads beyond the packed kernels' word budget).
__global__ __launch_bounds__(256) void gf_k_merge_find_bytes(const uint8_t* __restrict__ l_bases,
                                                             const uint8_t* __restrict__ l_quals,
                                                             const int64_t* __restrict__ l_off,
                                                             const uint8_t* __restrict__ r_bases,
                                                             const uint8_t* __restrict__ r_quals,
                                                             const int64_t* __restrict__ r_off, int64_t n,
                                                             int32_t* __restrict__ out_len,
                                                             int32_t* __restrict__ out_diff) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int len1 = (int)(l_off[p + 1] - l_off[p]), len2 = (int)(r_off[p + 1] - r_off[p]);
    int diff = 0;
    const int found = gf_merge_find_bytes(l_bases + l_off[p], l_quals + l_off[p], len1, r_bases + r_off[p],
                                          r_quals + r_off[p], len2, diff);
    out_len[p] = found ? len1 - found + len2 : 0;
    out_diff[p] = diff;
  }
}

// ---- K_merge_write: block per 256 pairs; each thread fetches its pair's layout, the merged
// pairs are listed in LDS and each is written by one wavefront, 64 bytes per step
// (read.rs:379-428).  in_len[p] = merged length from the find kernel (0 = not merged),
// out_pos[p] = where it goes. ----
__global__ __launch_bounds__(256) void gf_k_merge_write(const uint8_t* __restrict__ l_bases,
                                                        const uint8_t* __restrict__ l_quals,
                                                        const int64_t* __restrict__ l_off,
                                                        const uint8_t* __restrict__ r_bases,
                                                        const uint8_t* __restrict__ r_quals,
                                                        const int64_t* __restrict__ r_off, int64_t n,
                                                        const int32_t* __restrict__ in_len,
                                                        const int64_t* __restrict__ out_pos,
                                                        uint8_t* __restrict__ out_bases,
                                                        uint8_t* __restrict__ out_quals) {
  __shared__ unsigned int s_cnt;
  __shared__ int64_t s_l[256], s_r[256], s_dst[256];
  __shared__ int s_len1[256], s_len2[256], s_mlen[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
    __syncthreads();
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t p0 = base + threadIdx.x;
    const int ml = p0 < n ? in_len[p0] : 0;
    const bool merged = ml > 0;
    const unsigned int slot = gf_wave_append_lds(merged, &s_cnt);
    if (merged) {
      const int64_t lo = l_off[p0], ro = r_off[p0];
      s_l[slot] = lo;
      s_r[slot] = ro;
      s_len1[slot] = (int)(l_off[p0 + 1] - lo);
      s_len2[slot] = (int)(r_off[p0 + 1] - ro);
      s_mlen[slot] = ml;
      s_dst[slot] = out_pos[p0];
    }
    __syncthreads();
    const unsigned int cnt = s_cnt;
    for (unsigned int e = wave; e < cnt; e += 4) {
      const int len1 = s_len1[e], len2 = s_len2[e], mlen = s_mlen[e];
      const int offset = mlen - len2, olen = len1 - offset;
      const uint8_t* s1 = l_bases + s_l[e];
      const uint8_t* q1 = l_quals + s_l[e];
      const uint8_t* s2 = r_bases + s_r[e];
      const uint8_t* q2 = r_quals + s_r[e];
      uint8_t* os = out_bases + s_dst[e];
      uint8_t* oq = out_quals + s_dst[e];
      for (int k0 = 0; k0 < mlen; k0 += 256) {  // four 64-byte steps in flight
        uint8_t a1[4], b1[4], a2[4], b2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 64 * u + lane;
          a1[u] = b1[u] = a2[u] = b2[u] = 0;
          if (k < mlen) {
            if (k < len1) { a1[u] = s1[k]; b1[u] = q1[k]; }
            if (k >= offset) { a2[u] = s2[len2 - 1 - (k - offset)]; b2[u] = q2[len2 - 1 - (k - offset)]; }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 64 * u + lane;
          if (k < mlen) {
            uint8_t cs = a1[u], cq = b1[u];
            if (k >= offset) {
              cs = gf_complement(a2[u]);
              cq = b2[u];
              if (k - offset < olen) {
                if (a1[u] != cs) {
                  if (b1[u] >= '?' && cq <= '0') { cs = a1[u]; cq = b1[u]; }
                } else {
                  const uint32_t q = (uint32_t)b1[u] + (uint32_t)cq - 33u;  // add the pair's qualities, cap at 'Z'
                  cq = q >= (uint32_t)'Z' ? (uint8_t)'Z' : (uint8_t)q;
                }
              }
            }
            os[k] = cs;
            oq[k] = cq;
          }
        }
      }
    }
  }
}
