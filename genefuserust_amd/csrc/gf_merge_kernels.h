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
//   gf_k_merge_find_stream  thread per pair; each wavefront packs the spans of its next 64
//                   pairs into LDS tiles, then every lane slides R1's packed words past the
//                   first 16 bases of rc(R2); candidates with <= 2 mismatches there get the
//                   full-overlap count (still packed) and, with <= 2 in total, the quality test
//   gf_k_merge_write wave per merged pair: assembles the merged read and its quality
// Pairs the packed form cannot decide exactly (a read longer than the kernel's word
// budget, or both reads holding a byte outside A/C/G/T where
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

// even-bit mask of the bases below L in word j of a read
__device__ __forceinline__ uint32_t gf_len_mask2(int L, int j) {
  const int k = L - 16 * j;
  return k >= 16 ? 0x55555555u : (k <= 0 ? 0u : (((1u << (2 * k)) - 1u) & 0x55555555u));
}

// ---- K_merge_find_stream: thread per pair, packing folded in ----
// As in gf_k_seedverify_stream, consecutive reads are consecutive bytes: each wavefront
// converts the spans of its next 64 pairs — one of R1, one of R2 — to the packed form with
// coalesced, non-temporal 16-byte loads straight into two LDS tiles of its own, and every
// lane cuts its pair's words out of them.  rc(R2) is read off the forward tile: word j of
// the reverse complement is the field-reversed, complemented word that ends 16j bases
// before the end of R2.  No packed streams in HBM, no packing kernels.

// like gf_convert16, but valid = ACGTacgt (the bytes whose reverse complement is not 'N')
__device__ __forceinline__ void gf_convert16_nocase(const uint4& q, uint32_t& code32, uint32_t& bad16) {
  // (as gf_convert16: the doubled codes where they stand serve the table look-up and, by dot products, the packing)
  const uint32_t z0 = q.x & 0x06060606u, z1 = q.y & 0x06060606u, z2 = q.z & 0x06060606u, z3 = q.w & 0x06060606u;
  const uint32_t d0 = (q.x & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z0);
  const uint32_t d1 = (q.y & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z1);
  const uint32_t d2 = (q.z & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z2);
  const uint32_t d3 = (q.w & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z3);
  const uint32_t e0 = __builtin_amdgcn_udot4(z0, 0x40100401u, 0u, false), e1 = __builtin_amdgcn_udot4(z1, 0x40100401u, 0u, false);
  const uint32_t e2 = __builtin_amdgcn_udot4(z2, 0x40100401u, 0u, false), e3 = __builtin_amdgcn_udot4(z3, 0x40100401u, 0u, false);
  code32 = ((e0 + (e1 << 8) + (e2 << 16)) >> 1) | ((e3 >> 1) << 24);
  bad16 = 0;
  if (d0 | d1 | d2 | d3) {
    const uint32_t n0 = (((d0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d0) & 0x80808080u;
    const uint32_t n1 = (((d1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d1) & 0x80808080u;
    const uint32_t n2 = (((d2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d2) & 0x80808080u;
    const uint32_t n3 = (((d3 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d3) & 0x80808080u;
    bad16 = (((n0 >> 7) * 0x00204081u) >> 21 & 0xFu) | ((((n1 >> 7) * 0x00204081u) >> 21 & 0xFu) << 4) |
            ((((n2 >> 7) * 0x00204081u) >> 21 & 0xFu) << 8) | ((((n3 >> 7) * 0x00204081u) >> 21 & 0xFu) << 12);
  }
}

// 16 2-bit fields of x in reverse order
__device__ __forceinline__ uint32_t gf_field_reverse_dev(uint32_t x) {
  x = __brev(x);
  return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// Word j of rc(R2) from the forward tile: bases [end - 16(j+1), end - 16j) of the stream,
// reversed and complemented; `end` = stream position one past R2's last base.  The tile
// keeps 16 bases of slack in front (data starts at stream position 16), so the position
// never goes negative; bases before R2's first are masked by the caller's length masks.
__device__ __forceinline__ uint32_t gf_rc_word(const uint32_t* s_pk, uint32_t end, int j) {
  const uint32_t p = end - 16u * (uint32_t)(j + 1);
  const uint32_t w = __builtin_amdgcn_alignbit(s_pk[(p >> 4) + 1], s_pk[p >> 4], 2u * (p & 15u));
  return gf_field_reverse_dev(w) ^ 0xAAAAAAAAu;
}
// the same for the "unusable base" flags (1 bit per base in the tile), returned at the even bits
__device__ __forceinline__ uint32_t gf_rc_flags(const uint32_t* s_iv, uint32_t end, int j) {
  const uint32_t p = end - 16u * (uint32_t)(j + 1);
  const uint32_t b = __builtin_amdgcn_alignbit(s_iv[(p >> 5) + 1], s_iv[p >> 5], p & 31u) & 0xFFFFu;
  return gf_spread16(__brev(b) >> 16);
}

template <int PW, bool NOCASE>
__device__ __forceinline__ void gf_stage_tile(const uint4* __restrict__ src, uint32_t chunks, uint32_t* s_pk,
                                              uint32_t* s_iv, int lane, int lead_chunks) {
  constexpr int TILE_CHUNKS = 64 * PW + 1;
  constexpr int NLOAD = (TILE_CHUNKS + 63) / 64;
  uint4 q[NLOAD];
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
    const gf_u32x4 t = __builtin_nontemporal_load((const gf_u32x4*)(src + (c < chunks ? c : chunks - 1)));
    q[k] = make_uint4(t.x, t.y, t.z, t.w);
  }
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
    uint32_t code32, bad16;
    if (NOCASE) gf_convert16_nocase(q[k], code32, bad16);
    else gf_convert16(q[k], code32, bad16);
    if (c < chunks) {
      s_pk[c + lead_chunks] = code32;
      ((uint16_t*)s_iv)[c + lead_chunks] = (uint16_t)bad16;
    }
  }
}

#ifndef GF_MF_WAVES_PER_SIMD
#define GF_MF_WAVES_PER_SIMD 4
#endif
template <int PW>
__global__ __launch_bounds__(256, PW <= 10 ? GF_MF_WAVES_PER_SIMD : 3) void gf_k_merge_find_stream(const uint8_t* __restrict__ l_bases,
                                                              const uint8_t* __restrict__ l_quals,
                                                              const int64_t* __restrict__ l_off,
                                                              const uint8_t* __restrict__ r_bases,
                                                              const uint8_t* __restrict__ r_quals,
                                                              const int64_t* __restrict__ r_off, int64_t n,
    const int64_t* __restrict__ l_qoff, const int64_t* __restrict__ r_qoff,  /* where a read's qualities start in l_quals / r_quals; null: at its bases' offset */
                                                              int32_t* __restrict__ out_len,
                                                              int32_t* __restrict__ out_diff) {
  constexpr int TILE_BYTES = 64 * 16 * PW;
  constexpr int TILE_CHUNKS = TILE_BYTES / 16 + 1;
  constexpr int PK_WORDS = TILE_CHUNKS + PW + 4;           // +1 chunk of slack in front of R2's tile
  constexpr int IV_WORDS = (TILE_CHUNKS + PW + 4) / 2 + 2;
  __shared__ uint32_t s_pk1_all[4][PK_WORDS], s_pk2_all[4][PK_WORDS];
  __shared__ uint32_t s_iv1_all[4][IV_WORDS], s_iv2_all[4][IV_WORDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* s_pk1 = s_pk1_all[wave];
  uint32_t* s_pk2 = s_pk2_all[wave];
  uint32_t* s_iv1 = s_iv1_all[wave];
  uint32_t* s_iv2 = s_iv2_all[wave];
  if (lane == 0) {  // the slack in front of R2's tile: never used unmasked, but never garbage either
    s_pk2[0] = 0;
    s_iv2[0] = 0;
  }
  const int64_t n_groups = (n + 63) / 64;
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < n_groups; g += (int64_t)gridDim.x * 4) {
    const int64_t g0 = g * 64, g1 = g0 + 64 < n ? g0 + 64 : n;
    int64_t p0 = g0;
    while (p0 < g1) {
      // the pairs p0 .. p0+nfit-1 (a prefix of the group) fit in both tiles
      const int64_t base1 = l_off[p0], base2 = r_off[p0];
      const uint8_t* q01 = l_bases + base1;
      const uint8_t* q02 = r_bases + base2;
      const uint32_t mis1 = (uint32_t)((uintptr_t)q01 & 15u), mis2 = (uint32_t)((uintptr_t)q02 & 15u);
      const int64_t p = p0 + lane;
      int64_t lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
      if (p < g1) {
        lo1 = l_off[p]; hi1 = l_off[p + 1];
        lo2 = r_off[p]; hi2 = r_off[p + 1];
      }
      const bool fits = p < g1 && (uint64_t)(hi1 - base1) + mis1 <= (uint64_t)TILE_BYTES &&
                        (uint64_t)(hi2 - base2) + mis2 <= (uint64_t)TILE_BYTES;
      int nfit = __popcll(__ballot(fits));
      const bool oversize = nfit == 0;  // a read larger than a tile: the byte loop for this pair
      if (oversize) nfit = 1;
      gf_wave_lds_sync();  // the previous tile's LDS reads are done
      if (!oversize) {
        const uint32_t chunks1 = (uint32_t)((l_off[p0 + nfit] - base1) + mis1 + 15) >> 4;
        const uint32_t chunks2 = (uint32_t)((r_off[p0 + nfit] - base2) + mis2 + 15) >> 4;
        if (chunks1) gf_stage_tile<PW, false>((const uint4*)(q01 - mis1), chunks1, s_pk1, s_iv1, lane, 0);
        __builtin_amdgcn_sched_barrier(0);  // one tile's loads in registers at a time
        if (chunks2) gf_stage_tile<PW, true>((const uint4*)(q02 - mis2), chunks2, s_pk2, s_iv2, lane, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      gf_wave_lds_sync();
      if (lane < nfit) {
        const int64_t len1_64 = hi1 - lo1, len2_64 = hi2 - lo2;
        const int64_t lim64 = len1_64 < len2_64 ? len1_64 : len2_64;
        int found = 0, diff = 0;
        if (lim64 >= GF_MERGE_MIN_OVERLAP) {
          const uint8_t* s1 = l_bases + lo1;
          const uint8_t* q1 = l_quals + (l_qoff ? l_qoff[p] : lo1);
          const uint8_t* s2 = r_bases + lo2;
          const uint8_t* q2 = r_quals + (r_qoff ? r_qoff[p] : lo2);
          bool bytes_path = oversize || len1_64 > 16 * PW || len2_64 > 16 * PW;
          const int len1 = (int)len1_64, len2 = (int)len2_64;
          if (!bytes_path) {
            const uint32_t pos1 = (uint32_t)(lo1 - base1) + mis1;
            const uint32_t end2 = (uint32_t)(hi2 - base2) + mis2 + 16u;  // one past R2's last base (+ the slack chunk)
            const uint32_t w1 = pos1 >> 4, sh1 = 2u * (pos1 & 15u);
            uint32_t A[PW], IA[PW];
            uint32_t anyA = 0, anyB = 0;
#pragma unroll
            for (int j = 0; j < PW; ++j) {
              A[j] = gf_cut_pk(s_pk1, w1, sh1, j);
              IA[j] = gf_cut_iv(s_iv1, pos1, len1, j);
              anyA |= IA[j] & gf_len_mask2(len1, j);
              if (16 * j < len2) anyB |= gf_rc_flags(s_iv2, end2, j) & gf_len_mask2(len2, j);
            }
            // A byte outside A/C/G/T in R1 and one outside ACGTacgt in R2 are EQUAL when R1's is a literal
            // 'N' (the reverse complement of R2's is 'N', sequence.rs:51-59) and a mismatch otherwise.  The
            // packed form only knows "unusable", so: the 16-column pre-test counts a column where R2 is
            // unusable as no mismatch (it may only under-count: nothing is rejected wrongly), and the full
            // count looks at R1's byte for the columns where both are unusable.  (The first form sent every
            // such pair through the byte loop: with 0.1 % N in the genes nearly every wavefront had one.)
            const bool both_bad = anyA && anyB;
            {
              const int lim = len1 < len2 ? len1 : len2;
              const int o_hi = len1 - GF_MERGE_MIN_OVERLAP, o_lo = len1 - lim;  // offsets to try, high to low
              const uint32_t b0 = gf_rc_word(s_pk2, end2, 0), ib0 = gf_rc_flags(s_iv2, end2, 0);
              const uint32_t pre_mask = 0x55555555u & ~ib0;
              bool done = false;
              // the full test of one offset (rare: a random overlap passes the 16-column pre-test once in a million)
              auto full_test = [&](int o) {
                  // all columns of this overlap, 16 per word, R1's words cut from the tile
                  const int olen = len1 - o;
                  int cnt = 0, c0 = -1, c1 = -1;
#pragma unroll
                  for (int jj = 0; jj < PW; ++jj) {
                    const uint32_t cm = gf_len_mask2(olen, jj);
                    if (cm) {
                      const uint32_t pa = pos1 + (uint32_t)o + 16u * (uint32_t)jj;
                      const uint32_t ww = __builtin_amdgcn_alignbit(s_pk1[(pa >> 4) + 1], s_pk1[pa >> 4], 2u * (pa & 15u));
                      const uint32_t fa = __builtin_amdgcn_alignbit(s_iv1[(pa >> 5) + 1], s_iv1[pa >> 5], pa & 31u) & 0xFFFFu;
                      const uint32_t xx = ww ^ gf_rc_word(s_pk2, end2, jj);
                      const uint32_t fa2 = gf_spread16(fa), fb2 = gf_rc_flags(s_iv2, end2, jj);
                      uint32_t mm = ((xx | (xx >> 1)) | fa2 | fb2) & cm;
                      if (both_bad) {  // both unusable: equal iff R1's byte is a literal 'N'
                        uint32_t bb = fa2 & fb2 & cm;
                        while (bb) {
                          const int bit = __builtin_ctz(bb);
                          if (s1[o + 16 * jj + (bit >> 1)] == 'N') mm &= ~(1u << bit);
                          bb &= bb - 1;
                        }
                      }
                      cnt += __popc(mm);
                      while (mm && c1 < 0) {
                        const int col = 16 * jj + (__builtin_ctz(mm) >> 1);
                        if (c0 < 0) c0 = col; else c1 = col;
                        mm &= mm - 1;
                      }
                    }
                  }
                  if (cnt > 2) return;
                  bool ok = true;
                  if (c0 >= 0) ok = gf_lowq_pair(q1[o + c0], q2[len2 - 1 - c0]);
                  if (ok && c1 >= 0) ok = gf_lowq_pair(q1[o + c1], q2[len2 - 1 - c1]);
                  if (ok) {
                    found = olen;
                    diff = cnt;
                    done = true;
                  }
              };
              // Offsets high to low (the smallest overlap first, read.rs:339).  Per word of R1 the 16 pre-tests are
              // straight-line code with compile-time shifts — eight instructions each, no loop control — and their
              // outcomes are gathered in a bit mask; the offsets outside [o_lo, o_hi] are masked out afterwards and
              // the few survivors take the full test.  (As a runtime loop over the shifts with the range checks
              // inside, the pre-tests were two thirds of the kernel's instructions.)
              const bool wave_bad = __ballot(anyA != 0) != 0;
#pragma unroll
              for (int j = PW - 1; j >= 0; --j) {
                const uint32_t alo = A[j], ahi = j + 1 < PW ? A[j + 1] : 0u;
                const uint32_t ilo = IA[j], ihi = j + 1 < PW ? IA[j + 1] : 0x55555555u;
                const int hi_s = o_hi - 16 * j, lo_s = o_lo - 16 * j;  // this word's offsets: s in [lo_s, hi_s]
                uint32_t vm = 0;
                if (!done && hi_s >= 0 && lo_s <= 15) {
                  const int sa = lo_s < 0 ? 0 : lo_s, sb = hi_s > 15 ? 15 : hi_s;
                  vm = ((2u << sb) - 1u) & ~((1u << sa) - 1u);
                }
                if (__ballot(vm != 0) == 0) continue;  // (wave-uniform) nobody has an offset in this word
                uint32_t cmask = 0;
#pragma unroll
                for (int s = 15; s >= 0; --s) {
                  // first 16 columns: R1 bases o..o+15 against rc(R2) bases 0..15
                  uint32_t t = __builtin_amdgcn_alignbit(ahi, alo, 2u * (uint32_t)s) ^ b0;
                  t |= t >> 1;
                  if (wave_bad) t |= __builtin_amdgcn_alignbit(ihi, ilo, 2u * (uint32_t)s);
                  cmask = (cmask << 1) | (__popc(t & pre_mask) <= 2 ? 1u : 0u);
                }
                cmask &= vm;  // (bit s = offset 16 j + s)
                while (cmask != 0 && !done) {
                  const int s = 31 - __builtin_clz(cmask);
                  cmask &= ~(1u << s);
                  full_test(16 * j + s);
                }
              }
            }
          }
          if (bytes_path) found = gf_merge_find_bytes(s1, q1, len1, s2, q2, len2, diff);
          out_len[p] = found ? len1 - found + len2 : 0;
        } else {
          out_len[p] = 0;
        }
        out_diff[p] = diff;
      }
      p0 += nfit;
    }
  }
}

// Every pair through the byte loop (reads beyond the packed kernels' word budget).
__global__ __launch_bounds__(256) void gf_k_merge_find_bytes(const uint8_t* __restrict__ l_bases,
                                                             const uint8_t* __restrict__ l_quals,
                                                             const int64_t* __restrict__ l_off,
                                                             const uint8_t* __restrict__ r_bases,
                                                             const uint8_t* __restrict__ r_quals,
                                                             const int64_t* __restrict__ r_off, int64_t n,
    const int64_t* __restrict__ l_qoff, const int64_t* __restrict__ r_qoff,  /* where a read's qualities start in l_quals / r_quals; null: at its bases' offset */
                                                             int32_t* __restrict__ out_len,
                                                             int32_t* __restrict__ out_diff) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int len1 = (int)(l_off[p + 1] - l_off[p]), len2 = (int)(r_off[p + 1] - r_off[p]);
    int diff = 0;
    const int found = gf_merge_find_bytes(l_bases + l_off[p], l_quals + (l_qoff ? l_qoff[p] : l_off[p]), len1, r_bases + r_off[p],
                                          r_quals + (r_qoff ? r_qoff[p] : r_off[p]), len2, diff);
    out_len[p] = found ? len1 - found + len2 : 0;
    out_diff[p] = diff;
  }
}

// ---- K_merge_write: block per 256 pairs; each thread fetches its pair's layout, the merged
// pairs are listed in LDS and written by the wavefronts, two pairs at a time per wavefront
// and 320 bytes of each in flight: a merged pair is one load round trip, and the kernel's
// time is the number of such round trips a wavefront makes one after the other
// (read.rs:379-428).  in_len[p] = merged length from the find kernel (0 = not merged),
// out_pos[p] = where it goes. ----
struct GfMergeBytes {
  uint8_t a1[5], b1[5], a2[5], b2[5];  // R1 base/quality, R2 base/quality for output bytes lane, lane+64, ..
};

__device__ __forceinline__ void gf_mw_load(GfMergeBytes& g, const uint8_t* __restrict__ s1, const uint8_t* __restrict__ q1,
                                           const uint8_t* __restrict__ s2, const uint8_t* __restrict__ q2, int len1,
                                           int len2, int mlen, int offset, int k0, int lane) {
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int k = k0 + 64 * u + lane;
    g.a1[u] = g.b1[u] = g.a2[u] = g.b2[u] = 0;
    if (k < mlen) {
      if (k < len1) { g.a1[u] = s1[k]; g.b1[u] = q1[k]; }
      if (k >= offset) { g.a2[u] = s2[len2 - 1 - (k - offset)]; g.b2[u] = q2[len2 - 1 - (k - offset)]; }
    }
  }
}

__device__ __forceinline__ void gf_mw_store(const GfMergeBytes& g, uint8_t* __restrict__ os, uint8_t* __restrict__ oq,
                                            int len1, int mlen, int offset, int k0, int lane) {
  const int olen = len1 - offset;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int k = k0 + 64 * u + lane;
    if (k < mlen) {
      uint8_t cs = g.a1[u], cq = g.b1[u];
      if (k >= offset) {
        cs = gf_complement(g.a2[u]);
        cq = g.b2[u];
        if (k - offset < olen) {
          if (g.a1[u] != cs) {
            if (g.b1[u] >= '?' && cq <= '0') { cs = g.a1[u]; cq = g.b1[u]; }
          } else {
            const uint32_t q = (uint32_t)g.b1[u] + (uint32_t)cq - 33u;  // add the pair's qualities, cap at 'Z'
            cq = q >= (uint32_t)'Z' ? (uint8_t)'Z' : (uint8_t)q;
          }
        }
      }
      os[k] = cs;
      oq[k] = cq;
    }
  }
}

// Quality k of the merged read of a pair (read.rs:402-428), from the pair's own bytes: what gf_k_merge_write
// stores at out_quals[k].  Four loads at clamped positions and a selection — no branch, so that a caller's
// unrolled loop has all its loads in flight at once.
__device__ __forceinline__ uint8_t gf_merged_qual(const uint8_t* __restrict__ s1, const uint8_t* __restrict__ q1, int len1,
                                                  const uint8_t* __restrict__ s2, const uint8_t* __restrict__ q2, int len2,
                                                  int mlen, int k) {
  const int offset = mlen - len2;
  const int k1 = k < len1 ? k : len1 - 1;
  int j = len2 - 1 - (k - offset);
  j = j < 0 ? 0 : (j > len2 - 1 ? len2 - 1 : j);
  const uint8_t a1 = s1[k1], b1 = q1[k1], a2 = s2[j], b2 = q2[j];
  const uint32_t sum = (uint32_t)b1 + (uint32_t)b2 - 33u;
  const uint8_t same = sum >= (uint32_t)'Z' ? (uint8_t)'Z' : (uint8_t)sum;
  const uint8_t diff = (b1 >= '?' && b2 <= '0') ? b1 : b2;
  const uint8_t in_overlap = a1 != gf_complement(a2) ? diff : same;
  return k < offset ? b1 : (k >= len1 ? b2 : in_overlap);
}

__global__ __launch_bounds__(256) void gf_k_merge_write(const uint8_t* __restrict__ l_bases,
                                                        const uint8_t* __restrict__ l_quals,
                                                        const int64_t* __restrict__ l_off,
                                                        const uint8_t* __restrict__ r_bases,
                                                        const uint8_t* __restrict__ r_quals,
                                                        const int64_t* __restrict__ r_off, int64_t n,
    const int64_t* __restrict__ l_qoff, const int64_t* __restrict__ r_qoff,  /* where a read's qualities start in l_quals / r_quals; null: at its bases' offset */
                                                        const int32_t* __restrict__ in_len,
                                                        const int64_t* __restrict__ out_pos,
                                                        uint8_t* __restrict__ out_bases,
                                                        uint8_t* __restrict__ out_quals) {
  __shared__ unsigned int s_cnt;
  __shared__ int64_t s_l[256], s_r[256], s_dst[256], s_lq[256], s_rq[256];
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
      s_lq[slot] = l_qoff ? l_qoff[p0] : lo;
      s_rq[slot] = r_qoff ? r_qoff[p0] : ro;
      s_len1[slot] = (int)(l_off[p0 + 1] - lo);
      s_len2[slot] = (int)(r_off[p0 + 1] - ro);
      s_mlen[slot] = ml;
      s_dst[slot] = out_pos[p0];
    }
    __syncthreads();
    const unsigned int cnt = s_cnt;
    for (unsigned int e0 = 2 * wave; e0 < cnt; e0 += 8) {
      const unsigned int e1 = e0 + 1;
      const bool two = e1 < cnt;
      const unsigned int ex = two ? e1 : e0;
      const int len1a = s_len1[e0], len2a = s_len2[e0], mla = s_mlen[e0];
      const int len1b = s_len1[ex], len2b = s_len2[ex], mlb = two ? s_mlen[ex] : 0;
      const int offa = mla - len2a, offb = mlb - len2b;
      const int mmax = mla > mlb ? mla : mlb;
      for (int k0 = 0; k0 < mmax; k0 += 320) {
        GfMergeBytes ga, gb;
        gf_mw_load(ga, l_bases + s_l[e0], l_quals + s_lq[e0], r_bases + s_r[e0], r_quals + s_rq[e0], len1a, len2a, mla,
                   offa, k0, lane);
        gf_mw_load(gb, l_bases + s_l[ex], l_quals + s_lq[ex], r_bases + s_r[ex], r_quals + s_rq[ex], len1b, len2b, mlb,
                   offb, k0, lane);
        __builtin_amdgcn_sched_barrier(0);  // both pairs' loads before anybody's stores
        gf_mw_store(ga, out_bases + s_dst[e0], out_quals + s_dst[e0], len1a, mla, offa, k0, lane);
        gf_mw_store(gb, out_bases + s_dst[ex], out_quals + s_dst[ex], len1b, mlb, offb, k0, lane);
      }
    }
  }
}

// ---- K_merge_write_bases: the merged reads' bases alone, eight bytes per lane ----
// The pair pipeline (gf_scan_pairs_device) maps the merged reads and needs their qualities only for the few
// that are searched again or end in the hit list (gf_merged_qual).  Same block structure as gf_k_merge_write —
// the merged pairs of 256 are listed in LDS, a wavefront takes two of them at a time — but a lane produces eight
// consecutive bytes of the merged read: a piece of R1 as it is, a piece of rc(R2) from eight bytes of R2 turned
// round and complemented four at a time, a piece of the overlap from both (equal: done; different: the
// qualities of the differing columns decide, read.rs:402-428).  Only the pieces that straddle the start of
// rc(R2), the end of R1 or the end of the read go byte by byte.  A 300-base read is one round trip of 38 lanes
// where the byte-per-lane form makes five.
struct __attribute__((packed, aligned(1))) GfBytes8 { uint32_t v[2]; };

// reverse complement of four bases in a dword (sequence.rs:22-60: anything outside ACGTacgt -> 'N', upper case)
__device__ __forceinline__ uint32_t gf_rc4(uint32_t w) {
  const uint32_t y = (w >> 1) & 0x03030303u;
  const uint32_t d = (w & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x47544341u, y);  // 0 where the byte is ACGTacgt
  uint32_t c = __builtin_amdgcn_perm(0u, 0x43414754u, y);                              // A->T C->G T->A G->C
  if (d) {
    const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
    const uint32_t m = nz * 0xFFu;
    c = (c & ~m) | (0x4E4E4E4Eu & m);
  }
  return __builtin_bswap32(c);
}

// byte kk of the merged read
__device__ __forceinline__ uint8_t gf_merged_base(const uint8_t* __restrict__ s1, const uint8_t* __restrict__ q1, int len1,
                                                  const uint8_t* __restrict__ s2, const uint8_t* __restrict__ q2, int len2,
                                                  int offset, int kk) {
  if (kk < offset) return s1[kk];
  const int j = len2 - 1 - (kk - offset);
  uint8_t cs = gf_complement(s2[j]);
  if (kk < len1) {
    const uint8_t b1 = s1[kk];
    if (b1 != cs && q1[kk] >= '?' && q2[j] <= '0') cs = b1;
  }
  return cs;
}

// Bytes k .. k+7 of the merged read, no branches on where the piece lies: eight bytes of R1 at k and the eight
// bytes of R2 that rc(R2) takes its bytes k-offset .. k-offset+7 from are fetched whatever the piece covers
// (reading up to 7 bytes past either read's end: the 16-byte over-read every device buffer of the ABI allows; a
// piece that runs past the end of the merged read would start BEFORE R2 — it loads from R2's first byte and
// shifts), and every output byte picks its source by position.
__device__ __forceinline__ void gf_mwb_piece(const uint8_t* __restrict__ s1, const uint8_t* __restrict__ q1, int len1,
                                             const uint8_t* __restrict__ s2, const uint8_t* __restrict__ q2, int len2,
                                             int mlen, int offset, int k, uint8_t* __restrict__ os) {
  if (k >= mlen) return;
  const GfBytes8 a = *(const GfBytes8*)(s1 + (k < len1 ? k : 0));  // (not used at all when k >= len1)
  int jl = len2 - 1 - (k + 7 - offset);                            // R2 index of output byte k+7
  const int under = jl < 0 ? -jl : 0;                              // > 0 only in the read's last piece
  if (k + 8 <= offset) jl = 0;  // a piece of R1 alone: rc(R2) is not looked at (and jl would lie past R2's end)
  const GfBytes8 t = *(const GfBytes8*)(s2 + (jl < 0 ? 0 : jl));
  uint64_t t64 = (uint64_t)t.v[0] | ((uint64_t)t.v[1] << 32);
  t64 <<= 8 * under;
  uint32_t o0 = gf_rc4((uint32_t)(t64 >> 32)), o1 = gf_rc4((uint32_t)t64);  // rc(R2) bytes for k..k+3, k+4..k+7
  // bytes below `offset` come from R1
  const int n1 = offset - k;  // number of leading bytes that are R1's
  if (n1 > 0) {
    const uint32_t m0 = n1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n1)) - 1u);
    const uint32_t m1 = n1 >= 8 ? 0xFFFFFFFFu : (n1 <= 4 ? 0u : ((1u << (8 * (n1 - 4))) - 1u));
    o0 = (a.v[0] & m0) | (o0 & ~m0);
    o1 = (a.v[1] & m1) | (o1 & ~m1);
  }
  // inside the overlap (offset <= pos < len1) a column where the reads disagree is R1's when R1 is sure and R2 is not
  if (k < len1 && k + 8 > offset) {
    uint32_t x0 = a.v[0] ^ o0, x1 = a.v[1] ^ o1;
    if (x0 | x1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = k + i;
        const uint32_t xb = ((i < 4 ? x0 : x1) >> (8 * (i & 3))) & 0xFFu;
        if (xb && pos >= offset && pos < len1 && pos < mlen) {
          if (q1[pos] >= '?' && q2[len2 - 1 - (pos - offset)] <= '0') {
            const uint32_t b1 = ((i < 4 ? a.v[0] : a.v[1]) >> (8 * (i & 3))) & 0xFFu;
            if (i < 4) o0 = (o0 & ~(0xFFu << (8 * (i & 3)))) | (b1 << (8 * (i & 3)));
            else o1 = (o1 & ~(0xFFu << (8 * (i & 3)))) | (b1 << (8 * (i & 3)));
          }
        }
      }
    }
  }
  if (k + 8 <= mlen) {
    GfBytes8 o;
    o.v[0] = o0;
    o.v[1] = o1;
    *(GfBytes8*)(os + k) = o;
  } else {
    const uint64_t o64 = (uint64_t)o0 | ((uint64_t)o1 << 32);
    for (int i = 0; k + i < mlen; ++i) os[k + i] = (uint8_t)(o64 >> (8 * i));
  }
}

// The loads of a piece (gf_mwb_piece's first half), so that a thread can have several pieces' loads in flight
// before it looks at any of them.
struct GfMwbLoad {
  GfBytes8 a, t;
  int under;
};
__device__ __forceinline__ GfMwbLoad gf_mwb_fetch(const uint8_t* __restrict__ s1, int len1, const uint8_t* __restrict__ s2,
                                                   int len2, int offset, int k) {
  GfMwbLoad L;
  L.a = *(const GfBytes8*)(s1 + (k < len1 ? k : 0));  // (not used at all when k >= len1)
  int jl = len2 - 1 - (k + 7 - offset);               // R2 index of output byte k+7
  L.under = jl < 0 ? -jl : 0;                         // > 0 only in the read's last piece
  if (k + 8 <= offset) jl = 0;  // a piece of R1 alone: rc(R2) is not looked at (and jl would lie past R2's end)
  L.t = *(const GfBytes8*)(s2 + (jl < 0 ? 0 : jl));
  return L;
}
// ... and the rest of gf_mwb_piece on the bytes fetched
__device__ __forceinline__ void gf_mwb_finish(const GfMwbLoad& L, const uint8_t* __restrict__ q1, int len1,
                                              const uint8_t* __restrict__ q2, int len2, int mlen, int offset, int k,
                                              uint8_t* __restrict__ os) {
  uint64_t t64 = (uint64_t)L.t.v[0] | ((uint64_t)L.t.v[1] << 32);
  t64 <<= 8 * L.under;
  uint32_t o0 = gf_rc4((uint32_t)(t64 >> 32)), o1 = gf_rc4((uint32_t)t64);  // rc(R2) bytes for k..k+3, k+4..k+7
  const int n1 = offset - k;  // number of leading bytes that are R1's
  if (n1 > 0) {
    const uint32_t m0 = n1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n1)) - 1u);
    const uint32_t m1 = n1 >= 8 ? 0xFFFFFFFFu : (n1 <= 4 ? 0u : ((1u << (8 * (n1 - 4))) - 1u));
    o0 = (L.a.v[0] & m0) | (o0 & ~m0);
    o1 = (L.a.v[1] & m1) | (o1 & ~m1);
  }
  // inside the overlap (offset <= pos < len1) a column where the reads disagree is R1's when R1 is sure and R2 is not
  if (k < len1 && k + 8 > offset) {
    const uint32_t x0 = L.a.v[0] ^ o0, x1 = L.a.v[1] ^ o1;
    if (x0 | x1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = k + i;
        const uint32_t xb = ((i < 4 ? x0 : x1) >> (8 * (i & 3))) & 0xFFu;
        if (xb && pos >= offset && pos < len1 && pos < mlen) {
          if (q1[pos] >= '?' && q2[len2 - 1 - (pos - offset)] <= '0') {
            const uint32_t b1 = ((i < 4 ? L.a.v[0] : L.a.v[1]) >> (8 * (i & 3))) & 0xFFu;
            if (i < 4) o0 = (o0 & ~(0xFFu << (8 * (i & 3)))) | (b1 << (8 * (i & 3)));
            else o1 = (o1 & ~(0xFFu << (8 * (i & 3)))) | (b1 << (8 * (i & 3)));
          }
        }
      }
    }
  }
  if (k + 8 <= mlen) {
    GfBytes8 o;
    o.v[0] = o0;
    o.v[1] = o1;
    *(GfBytes8*)(os + k) = o;
  } else {
    const uint64_t o64 = (uint64_t)o0 | ((uint64_t)o1 << 32);
    for (int i = 0; k + i < mlen; ++i) os[k + i] = (uint8_t)(o64 >> (8 * i));
  }
}

// r03 b: the merged reads of a round of 256 pairs as ONE list of 8-byte pieces — a thread takes pieces t, t + 256, ..
// of the round, whatever reads they belong to, four of them in flight at a time.  (Two reads per wavefront, a piece per
// lane: 34 of 64 lanes had a piece of a 270-base read and one round trip's worth of loads in flight; 0.82 ms per 10 M
// pairs, its waves waiting 54 % of their cycles.)
__global__ __launch_bounds__(256) void gf_k_merge_write_bases(const uint8_t* __restrict__ l_bases,
                                                              const uint8_t* __restrict__ l_quals,
                                                              const int64_t* __restrict__ l_off,
                                                              const uint8_t* __restrict__ r_bases,
                                                              const uint8_t* __restrict__ r_quals,
                                                              const int64_t* __restrict__ r_off, int64_t n,
    const int64_t* __restrict__ l_qoff, const int64_t* __restrict__ r_qoff,  /* where a read's qualities start in l_quals / r_quals; null: at its bases' offset */
                                                              const int32_t* __restrict__ in_len,
                                                              const int64_t* __restrict__ out_pos,
                                                              uint8_t* __restrict__ out_bases) {
  __shared__ int s_wave[4];
  __shared__ int s_first[257];  // the pair's first piece in the round's list (pairs that did not merge: none)
  __shared__ int64_t s_l[256], s_r[256], s_dst[256], s_lq[256], s_rq[256];
  __shared__ int s_len1[256], s_len2[256], s_mlen[256];
  const int tid = threadIdx.x;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
    __syncthreads();  // (the previous round's readers are done with the arrays)
    const int64_t p0 = base + tid;
    const int ml = p0 < n ? in_len[p0] : 0;
    if (ml > 0) {
      const int64_t lo = l_off[p0], ro = r_off[p0];
      s_l[tid] = lo;
      s_r[tid] = ro;
      s_lq[tid] = l_qoff ? l_qoff[p0] : lo;
      s_rq[tid] = r_qoff ? r_qoff[p0] : ro;
      s_len1[tid] = (int)(l_off[p0 + 1] - lo);
      s_len2[tid] = (int)(r_off[p0 + 1] - ro);
      s_mlen[tid] = ml;
      s_dst[tid] = out_pos[p0];
    }
    int total;
    const int first = gf_block_exclusive_scan(ml > 0 ? (ml + 7) >> 3 : 0, s_wave, &total);
    s_first[tid] = first;
    if (tid == 0) s_first[256] = total;
    __syncthreads();
    for (int q0 = tid; q0 < total; q0 += 4 * 256) {
      GfMwbLoad L[4];
      int e[4], kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + 256 * u;
        e[u] = -1;
        if (q < total) {
          int i = 0, j = 256;  // the pair whose pieces hold q: the largest i with s_first[i] <= q
          while (j - i > 1) {
            const int mid = (i + j) >> 1;
            if (s_first[mid] <= q) i = mid; else j = mid;
          }
          e[u] = i;
          kk[u] = 8 * (q - s_first[i]);
          L[u] = gf_mwb_fetch(l_bases + s_l[i], s_len1[i], r_bases + s_r[i], s_len2[i], s_mlen[i] - s_len2[i], kk[u]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the four pieces' loads before anybody's stores
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e[u] >= 0) {
          const int i = e[u];
          gf_mwb_finish(L[u], l_quals + s_lq[i], s_len1[i], r_quals + s_rq[i], s_len2[i], s_mlen[i], s_mlen[i] - s_len2[i],
                        kk[u], out_bases + s_dst[i]);
        }
    }
  }
}
