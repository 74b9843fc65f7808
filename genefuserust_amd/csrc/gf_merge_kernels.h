// SURVEY.md §8(f)-2 — SequenceReadPair::fast_merge (src/core/read.rs:313-440) on the
// device: the step before the hot path.  Pairs are independent; one thread per pair.
//
// rc_right = reverse complement of R2 (anything outside ACGTacgt -> 'N', output upper case,
// sequence.rs:22-60), its quality reversed.  The smallest overlap olen >= 30 is taken
// for which every mismatch between R1's tail and rc_right's head is a "low quality"
// mismatch (one base >= Q30 i.e. >= '?', the other <= Q15 i.e. <= '0') and there are
// fewer than three of them (read.rs:339-367; the loop's `diff > low_qual_diff ||
// low_qual_diff >= 3` is order-independent: no high-quality mismatch, at most two
// low-quality ones).  merged = R1[0, len1-olen) + rc_right, overlap corrected (:402-428).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// out_pos[p] = where merged read p goes in out_bases/out_quals (caller-chosen);
// out_len[p] = merged length, 0 = the pair does not merge.  out_bases == nullptr is the
// sizing pass (lengths and diffs only), so that the caller can lay the merged reads out
// back to back — the layout gf_map_reads_device takes — with one prefix sum.
__global__ __launch_bounds__(256) void gf_k_fast_merge(const uint8_t* __restrict__ l_bases,
                                                       const uint8_t* __restrict__ l_quals,
                                                       const int64_t* __restrict__ l_off,
                                                       const uint8_t* __restrict__ r_bases,
                                                       const uint8_t* __restrict__ r_quals,
                                                       const int64_t* __restrict__ r_off, int64_t n,
                                                       const int64_t* __restrict__ out_pos,
                                                       uint8_t* __restrict__ out_bases,
                                                       uint8_t* __restrict__ out_quals, int32_t* __restrict__ out_len,
                                                       int32_t* __restrict__ out_diff) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int len1 = (int)(l_off[p + 1] - l_off[p]), len2 = (int)(r_off[p + 1] - r_off[p]);
    const uint8_t* s1 = l_bases + l_off[p];
    const uint8_t* q1 = l_quals + l_off[p];
    const uint8_t* s2 = r_bases + r_off[p];  // rc_right[i] = complement(s2[len2-1-i]), qual2[i] = q2[len2-1-i]
    const uint8_t* q2 = r_quals + r_off[p];
    const int lim = len1 < len2 ? len1 : len2;
    int found = 0, diff = 0;
    for (int olen = GF_MERGE_MIN_OVERLAP; olen <= lim; ++olen) {
      const int offset = len1 - olen;
      int d = 0;
      bool ok = true;
      for (int i = 0; i < olen; ++i) {
        if (s1[offset + i] != gf_complement(s2[len2 - 1 - i])) {
          const uint8_t a = q1[offset + i], b = q2[len2 - 1 - i];
          const bool lowq = (a >= '?' && b <= '0') || (a <= '0' && b >= '?');
          d += 1;
          if (!lowq || d >= 3) {
            ok = false;
            break;
          }
        }
      }
      if (ok) {
        found = olen;
        diff = d;
        break;
      }
    }
    if (!found) {
      out_len[p] = 0;
      out_diff[p] = 0;
      continue;
    }
    const int offset = len1 - found;
    out_len[p] = offset + len2;
    out_diff[p] = diff;
    if (!out_bases) continue;  // sizing pass: the caller turns out_len into out_pos
    uint8_t* os = out_bases + out_pos[p];
    uint8_t* oq = out_quals + out_pos[p];
    for (int i = 0; i < offset; ++i) {
      os[i] = s1[i];
      oq[i] = q1[i];
    }
    for (int i = 0; i < len2; ++i) {
      const uint8_t c2 = gf_complement(s2[len2 - 1 - i]), b = q2[len2 - 1 - i];
      uint8_t cs = c2, cq = b;
      if (i < found) {
        const uint8_t c1 = s1[offset + i], a = q1[offset + i];
        if (c1 != c2) {
          if (a >= '?' && b <= '0') { cs = c1; cq = a; }
        } else {
          const uint32_t q = (uint32_t)a + (uint32_t)b - 33u;  // add the pair's qualities, cap at 'Z'
          cq = q >= (uint32_t)'Z' ? (uint8_t)'Z' : (uint8_t)q;
        }
      }
      os[offset + i] = cs;
      oq[offset + i] = cq;
    }
  }
}
