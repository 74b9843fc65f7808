// The pair policy of PairEndScanner::scan_pair_end (src/core/pescanner.rs:427-518) for a whole pack
// of pairs, resident in HBM from the FASTQ records to the hit list:
//
//   merged = pair.fast_merge()                         gf_k_merge_find_stream / gf_k_merge_write
//   merged?  search the merged read only               one mapping pass over the merged slots
//   else     search R1, then R2                        one pass each over R1 / R2 in place, the
//                                                      pairs that merged skipped (GfTable::skip)
//   a read that mapped to two places (`mapable`, fusion_mapper.rs:107-115) in the wrong
//   direction (:118-123) is searched again as its reverse complement    gf_k_pair_classify,
//                                                      gf_k_pair_retry_write, a fourth (small) pass
//   matches are pushed per pair in the order merged | R1, R2            gf_k_pair_final_*
//
// Nothing goes back to the host in between: the kernels below select the candidates, build the
// reverse complements, and compact the matched reads (records + their bases and qualities, which
// the host-side tail FusionMapper::make_match / calc_distance needs) in the reference's push order,
// deterministically (two-level exclusive scans, no atomics).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"
#include "gf_compact_kernels.h"
#include "gf_merge_kernels.h"

#define GF_PTILE 256  // pairs per tile: one per thread, so that a wavefront reads and writes consecutive pairs
#define GF_PPER (GF_PTILE / GF_CTHREADS)
static_assert(GF_PTILE == GF_CTHREADS, "the pair kernels take one pair per thread");

#define GF_PS_NONE 0u
#define GF_PS_FWD 1u    // two segments in the required direction: a match on the read as it is
#define GF_PS_RETRY 2u  // two segments, wrong direction: its reverse complement is searched

struct GfPairIn {
  const uint8_t *l_bases, *l_quals, *r_bases, *r_quals;
  const int64_t *l_off, *r_off;
  // Where a read's qualities start in l_quals / r_quals; null: at its bases' offset (the layout gf_fastq_gather_device
  // writes).  Not null: the qualities were left where they are — l_quals is the FASTQ text itself and l_qoff[p]
  // the start of record p's quality line (gf_fastq_gather_lean_device) — and are fetched for the few reads that
  // need them: mismatching columns of an overlap, reverse-complement retries, hit records.
  const int64_t *l_qoff, *r_qoff;
  const uint8_t* m_bases;            // merged reads, back to back (their qualities: gf_pair_qual)
  const int64_t* m_off;              // int64[n+1]: an empty slot for a pair that did not merge
  const int32_t* m_len;              // 0 = not merged
  const int32_t* m_diff;
  const int32_t* m_rank;             // merged pair p is read m_rank[p] of the (compact) merged mapping pass
  const uint8_t *cM, *c1, *c2;       // counts of the three mapping passes
  const gf_seqmatch *mM, *m1, *m2;
  const uint8_t* gene_reversed;      // Fusion::is_reversed() per gene; null = all false
  int32_t n_genes;
};

// Indexer::in_required_direction (indexer.rs:541-608) for a two-segment mapping
__device__ __forceinline__ bool gf_dev_required_direction(const gf_seqmatch& a, const gf_seqmatch& b,
                                                          const uint8_t* __restrict__ rev, int n_genes) {
  const bool swap = a.seq_start > b.seq_start;
  const gf_seqmatch& left = swap ? b : a;
  const gf_seqmatch& right = swap ? a : b;
  if (left.position > 0 && right.position > 0) return true;
  if (left.position < 0 && right.position < 0) return false;
  const bool lrev = rev && left.contig >= 0 && left.contig < n_genes && rev[left.contig] != 0;
  const bool rrev = rev && right.contig >= 0 && right.contig < n_genes && rev[right.contig] != 0;
  if (lrev && !rrev) return false;
  if (!lrev && rrev) return true;
  if (left.contig < right.contig) return true;
  return false;  // (the reference's same-contig test compares left with itself, :598: never true)
}

// candidate s of pair p: 0 = merged read, 1 = R1, 2 = R2
__device__ __forceinline__ void gf_pair_candidate(const GfPairIn& P, int64_t p, int s, const uint8_t*& bases,
                                                  const uint8_t*& quals, int32_t& len, uint8_t& cnt,
                                                  const gf_seqmatch*& m) {
  if (s == 0) {
    const int64_t o = P.m_off[p];
    const int64_t j = P.m_len[p] > 0 ? (int64_t)P.m_rank[p] : 0;   // (cM / mM are indexed by merged read, not by pair)
    bases = P.m_bases + o; quals = nullptr; len = P.m_len[p]; cnt = P.m_len[p] > 0 ? P.cM[j] : (uint8_t)0; m = P.mM + 2 * j;
  } else if (s == 1) {
    const int64_t o = P.l_off[p];
    bases = P.l_bases + o; quals = P.l_qoff ? nullptr : P.l_quals + o; len = (int32_t)(P.l_off[p + 1] - o); cnt = P.c1[p]; m = P.m1 + 2 * p;
  } else {
    const int64_t o = P.r_off[p];
    bases = P.r_bases + o; quals = P.r_qoff ? nullptr : P.r_quals + o; len = (int32_t)(P.r_off[p + 1] - o); cnt = P.c2[p]; m = P.m2 + 2 * p;
  }
}

// The qualities of candidate s of pair p.  A merged read's qualities are not stored: the pipeline needs them for
// the reads that are searched again as reverse complements or end in the hit list — a few per thousand — and
// they follow from the pair's own bytes (read.rs:402-428).
struct GfPairQual {
  const uint8_t *q, *s1, *q1, *s2, *q2;
  int len1, len2, mlen;
  __device__ __forceinline__ uint8_t merged_at(int k) const { return gf_merged_qual(s1, q1, len1, s2, q2, len2, mlen, k); }
};
__device__ __forceinline__ GfPairQual gf_pair_qual(const GfPairIn& P, int64_t p, int s, const uint8_t* quals) {
  GfPairQual Q;
  Q.q = quals;
  Q.s1 = Q.q1 = Q.s2 = Q.q2 = nullptr;
  Q.len1 = Q.len2 = Q.mlen = 0;
  if (s == 0 && !quals) {
    const int64_t lo = P.l_off[p], ro = P.r_off[p];
    Q.s1 = P.l_bases + lo; Q.q1 = P.l_quals + (P.l_qoff ? P.l_qoff[p] : lo); Q.len1 = (int)(P.l_off[p + 1] - lo);
    Q.s2 = P.r_bases + ro; Q.q2 = P.r_quals + (P.r_qoff ? P.r_qoff[p] : ro); Q.len2 = (int)(P.r_off[p + 1] - ro);
    Q.mlen = P.m_len[p];
  } else if (s == 1 && !quals) {  // (qualities left in the text: gf_pair_candidate hands out no pointer)
    Q.q = P.l_quals + P.l_qoff[p];
  } else if (s == 2 && !quals) {
    Q.q = P.r_quals + P.r_qoff[p];
  }
  return Q;
}
__device__ __forceinline__ GfPairQual gf_pair_qual_none() {  // (a lane without a read to write)
  GfPairQual Q;
  Q.q = Q.s1 = Q.q1 = Q.s2 = Q.q2 = nullptr;
  Q.len1 = Q.len2 = Q.mlen = 0;
  return Q;
}

// The reads that the lanes in `mask` have to write, one after the other, every read by all 64 lanes of the
// wavefront (lane j: bytes j, j + 64, ..).  Hits and retries are a few per thousand pairs: a lane that copied
// its own read byte by byte was alone in its wavefront with one round trip per byte, and the kernel took as
// long as its longest read had bytes.  revcomp: the read's reverse complement (its qualities reversed).
__device__ __forceinline__ void gf_wave_write_reads(uint64_t mask, const uint8_t* b, const GfPairQual& Q, int len,
                                                    long long out, uint8_t* __restrict__ ob, uint8_t* __restrict__ oq,
                                                    bool revcomp) {
  const int lane = threadIdx.x & 63;
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    const uint8_t* bb = (const uint8_t*)__shfl((unsigned long long)b, l);
    GfPairQual R;
    R.q = (const uint8_t*)__shfl((unsigned long long)Q.q, l);
    R.s1 = (const uint8_t*)__shfl((unsigned long long)Q.s1, l);
    R.q1 = (const uint8_t*)__shfl((unsigned long long)Q.q1, l);
    R.s2 = (const uint8_t*)__shfl((unsigned long long)Q.s2, l);
    R.q2 = (const uint8_t*)__shfl((unsigned long long)Q.q2, l);
    R.len1 = __shfl(Q.len1, l);
    R.len2 = __shfl(Q.len2, l);
    R.mlen = __shfl(Q.mlen, l);
    const int ln = __shfl(len, l);
    const long long o = __shfl(out, l);
#pragma unroll 1
    for (int j = lane; j < ln; j += 64) {
      const int src = revcomp ? ln - 1 - j : j;
      const uint8_t base = bb[src];
      ob[o + j] = revcomp ? gf_complement(base) : base;
      oq[o + j] = R.q ? R.q[src] : R.merged_at(src);
    }
  }
}

__device__ __forceinline__ uint32_t gf_pair_status(const GfPairIn& P, int64_t p, int s, int32_t& len) {
  const uint8_t* b; const uint8_t* q; uint8_t cnt; const gf_seqmatch* m;
  gf_pair_candidate(P, p, s, b, q, len, cnt, m);
  if (cnt != 2) return GF_PS_NONE;  // mapping.len() < 2: not mapable (fusion_mapper.rs:107-115)
  return gf_dev_required_direction(m[0], m[1], P.gene_reversed, P.n_genes) ? GF_PS_FWD : GF_PS_RETRY;
}

// block-wide exclusive scan of two values at once (256 threads)
__device__ __forceinline__ void gf_block_scan2(int a, long long b, int* s_a, long long* s_b, int& ea, long long& eb,
                                               int& ta, long long& tb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int xa = a;
  long long xb = b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ya = __shfl_up(xa, o);
    const long long yb = __shfl_up(xb, o);
    if (lane >= o) { xa += ya; xb += yb; }
  }
  if (lane == 63) { s_a[wave] = xa; s_b[wave] = xb; }
  __syncthreads();
  int ba = 0; long long bb = 0;
  ta = 0; tb = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) { ba += s_a[w]; bb += s_b[w]; }
    ta += s_a[w]; tb += s_b[w];
  }
  ea = ba + xa - a;
  eb = bb + xb - b;
  __syncthreads();
}

// The merged reads' slots: exclusive prefix sums of the int32 lengths (bytes: m_off, int64[n+1]) and of the
// merged flags (reads: m_rank), tile sums first ...
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_len_tile_sums(const int32_t* __restrict__ len, int64_t n,
                                                                  uint32_t* __restrict__ tile_bytes,
                                                                  uint32_t* __restrict__ tile_reads) {
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  const int64_t r0 = (int64_t)blockIdx.x * GF_CTILE + (int64_t)threadIdx.x * GF_CPER;
  int c = 0;
  long long b = 0;
  for (int k = 0; k < GF_CPER; ++k)
    if (r0 + k < n && len[r0 + k] > 0) { c += 1; b += len[r0 + k]; }
  int ea, ta; long long eb, tb;
  gf_block_scan2(c, b, s_a, s_b, ea, eb, ta, tb);
  if (threadIdx.x == 0) {
    tile_reads[blockIdx.x] = (uint32_t)ta;
    tile_bytes[blockIdx.x] = (uint32_t)tb;
  }
}
// ... then (after gf_k_compact_scan of both tile arrays) the offsets: m_off[p] for every pair (an empty slot for
// a pair that did not merge), m_rank[p] and the compact c_off[m_rank[p]] = m_off[p] for the pairs that did.
// The merged mapping pass runs over c_off: the real reads first, empty reads behind them (gf_k_len_tail) —
// whole tiles of empty reads cost next to nothing, 84 % empty slots among the real ones cost a full pass.
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_len_offsets(const int32_t* __restrict__ len, int64_t n,
                                                                const int64_t* __restrict__ tile_off_bytes,
                                                                const int64_t* __restrict__ tile_off_reads,
                                                                const int64_t* __restrict__ d_total_bytes,
                                                                int64_t* __restrict__ offsets, int32_t* __restrict__ rank,
                                                                int64_t* __restrict__ c_off) {
  // (r03 b: the tile's lengths come in and its offsets go out through LDS, a wavefront's loads and stores contiguous —
  //  a thread owns 16 consecutive pairs, and taking them straight from memory put its lanes 64 and 128 bytes apart:
  //  0.20 ms per 10 M pairs for 40 MB in and 130 MB out)
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  __shared__ int s_len[GF_CTILE];
  __shared__ uint32_t s_rel[GF_CTILE];   // offset of pair i within the tile's bytes
  __shared__ uint32_t s_crel[GF_CTILE];  // the same for the tile's merged pairs, in their order
  const int64_t t0 = (int64_t)blockIdx.x * GF_CTILE;
  const int tid = threadIdx.x;
  for (int i = tid; i < GF_CTILE; i += GF_CTHREADS) s_len[i] = t0 + i < n ? len[t0 + i] : 0;
  __syncthreads();
  int c = 0;
  long long b = 0;
  for (int k = 0; k < GF_CPER; ++k) {
    const int l = s_len[tid * GF_CPER + k];
    if (l > 0) { c += 1; b += l; }
  }
  int ea, ta; long long eb, tb;
  gf_block_scan2(c, b, s_a, s_b, ea, eb, ta, tb);
  {
    uint32_t pos = (uint32_t)eb;
    int j = ea;
    for (int k = 0; k < GF_CPER; ++k) {
      const int i = tid * GF_CPER + k;
      s_rel[i] = pos;
      const int l = s_len[i];
      if (l > 0) {
        s_crel[j] = pos;
        s_len[i] = -(j + 1);  // (the length is spent: the pair's place among the tile's merged pairs, for its rank)
        pos += (uint32_t)l;
        j += 1;
      }
    }
  }
  __syncthreads();
  const int64_t base = tile_off_bytes[blockIdx.x], j0 = tile_off_reads[blockIdx.x];
  for (int i = tid; i < GF_CTILE; i += GF_CTHREADS) {
    const int64_t r = t0 + i;
    if (r >= n) break;
    offsets[r] = base + s_rel[i];
    if (s_len[i] < 0) rank[r] = (int32_t)(j0 + (-s_len[i] - 1));
  }
  for (int j = tid; j < ta; j += GF_CTHREADS) c_off[j0 + j] = base + s_crel[j];
  if (blockIdx.x == 0 && tid == 0) offsets[n] = *d_total_bytes;
}
// the empty reads behind the merged ones: c_off[k] = total bytes for k = number of merged pairs .. n
__global__ void gf_k_len_tail(const int64_t* __restrict__ d_n_merged, const int64_t* __restrict__ d_total_bytes, int64_t n,
                              int64_t* __restrict__ c_off) {
  const int64_t nm = *d_n_merged, tot = *d_total_bytes;
  for (int64_t k = nm + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (int64_t)gridDim.x * blockDim.x)
    c_off[k] = tot;
}

// ---- classify: which candidates matched as they are, which are searched again reversed ----
// tile_rc / tile_rb: retries (reads / bytes) per tile.
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_pair_classify(GfPairIn P, int64_t n, uint8_t* __restrict__ st,
                                                                  uint32_t* __restrict__ tile_rc,
                                                                  uint32_t* __restrict__ tile_rb) {
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  const int64_t p0 = (int64_t)blockIdx.x * GF_PTILE + (int64_t)threadIdx.x * GF_PPER;
  int rc = 0;
  long long rb = 0;
  for (int k = 0; k < GF_PPER; ++k) {
    const int64_t p = p0 + k;
    if (p >= n) break;
    const bool is_merged = P.m_len[p] > 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      uint32_t v = GF_PS_NONE;
      int32_t len = 0;
      if (is_merged ? s == 0 : s != 0) v = gf_pair_status(P, p, s, len);
      st[3 * p + s] = (uint8_t)v;
      if (v == GF_PS_RETRY) { rc += 1; rb += len; }
    }
  }
  int ea, ta; long long eb, tb;
  gf_block_scan2(rc, rb, s_a, s_b, ea, eb, ta, tb);
  if (threadIdx.x == 0) {
    tile_rc[blockIdx.x] = (uint32_t)ta;
    tile_rb[blockIdx.x] = (uint32_t)tb;
  }
}

// reverse complement of a read as SequenceRead::reverse_complement builds it (read.rs:243-261 over
// sequence.rs:22-60): bases complemented to UPPER case, anything but ACGTacgt -> N, qualities reversed
__device__ __forceinline__ uint8_t gf_complement_base(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

// ---- retry_write: the reverse complements of the retried reads, back to back, in candidate order ----
// slot_of[3p+s] = index of the candidate in the retry batch.  Retries beyond the capacities are
// dropped and reported (totals: overflow), never half-written.
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_pair_retry_write(
    GfPairIn P, int64_t n, const uint8_t* __restrict__ st, const int64_t* __restrict__ tile_off_rc,
    const int64_t* __restrict__ tile_off_rb, int64_t cap_reads, int64_t cap_bytes, int64_t* __restrict__ r_off,
    uint8_t* __restrict__ r_bases, uint8_t* __restrict__ r_quals, int32_t* __restrict__ slot_of) {
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  const int64_t p0 = (int64_t)blockIdx.x * GF_PTILE + (int64_t)threadIdx.x * GF_PPER;
  int rc = 0;
  long long rb = 0;
  for (int k = 0; k < GF_PPER; ++k) {
    const int64_t p = p0 + k;
    if (p >= n) break;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (st[3 * p + s] == GF_PS_RETRY) {
        const uint8_t* b; const uint8_t* q; int32_t len; uint8_t cnt; const gf_seqmatch* m;
        gf_pair_candidate(P, p, s, b, q, len, cnt, m);
        rc += 1;
        rb += len;
      }
  }
  int ea, ta; long long eb, tb;
  gf_block_scan2(rc, rb, s_a, s_b, ea, eb, ta, tb);
  if (__ballot(rc != 0) == 0) return;  // (whole wavefronts: the reads are written by all 64 lanes)
  int64_t k_out = tile_off_rc[blockIdx.x] + ea;
  int64_t b_out = tile_off_rb[blockIdx.x] + eb;
  const int64_t p = p0;  // GF_PPER == 1
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bool mine = p < n && st[3 * p + s] == GF_PS_RETRY;
    const uint8_t* b = nullptr; const uint8_t* q = nullptr; int32_t len = 0; uint8_t cnt; const gf_seqmatch* m;
    bool fits = false;
    GfPairQual Q = gf_pair_qual_none();
    if (mine) {
      gf_pair_candidate(P, p, s, b, q, len, cnt, m);
      fits = k_out < cap_reads && b_out + len <= cap_bytes;
      slot_of[3 * p + s] = fits ? (int32_t)k_out : -1;
      if (fits) {
        r_off[k_out] = b_out;
        Q = gf_pair_qual(P, p, s, q);
      }
    }
    gf_wave_write_reads(__ballot(fits), b, Q, len, (long long)b_out, r_bases, r_quals, true);
    if (mine) {
      k_out += 1;
      b_out += len;
    }
  }
}

// offsets of the unused slots of the retry batch (empty reads at the end of the buffer), and the
// overflow flag.  totals: [0] hits, [1] hit bytes, [2] merged pairs, [3] retried reads, [4] overflow bits
__global__ void gf_k_pair_retry_tail(const int64_t* __restrict__ d_n_retry, const int64_t* __restrict__ d_retry_bytes,
                                     int64_t cap_reads, int64_t cap_bytes, int64_t* __restrict__ r_off,
                                     int64_t* __restrict__ totals, unsigned int* __restrict__ n_exact) {
  const int64_t nr = *d_n_retry, nb = *d_retry_bytes;
  // Over capacity: the whole retry batch is emptied (every offset 0) and the flag raised — the caller
  // runs the pack again with room for all (gf_scan_pairs_device: totals[4]); a partly searched batch
  // would look like a result.
  const bool over = nr > cap_reads || nb > cap_bytes;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= cap_reads; k += (int64_t)gridDim.x * blockDim.x) {
    if (over) r_off[k] = 0;
    else if (k >= nr) r_off[k] = nb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    totals[3] = nr;
    if (over) totals[4] |= 1;
    *n_exact = over ? 0u : (unsigned int)nr;  // the reads the retry pass maps (it takes them one wavefront each)
  }
}

// ---- final: the matches of the pack, in the reference's push order ----
__device__ __forceinline__ bool gf_pair_final(const GfPairIn& P, int64_t p, int s, const uint8_t* __restrict__ st,
                                              const int32_t* __restrict__ slot_of, const uint8_t* __restrict__ cR,
                                              const gf_seqmatch* __restrict__ mR, int& rc_slot) {
  const uint8_t v = st[3 * p + s];
  rc_slot = -1;
  if (v == GF_PS_FWD) return true;
  if (v != GF_PS_RETRY) return false;
  const int32_t k = slot_of[3 * p + s];
  if (k < 0 || cR[k] != 2) return false;
  if (!gf_dev_required_direction(mR[2 * (int64_t)k], mR[2 * (int64_t)k + 1], P.gene_reversed, P.n_genes)) return false;
  rc_slot = k;
  return true;
}

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_pair_final_count(
    GfPairIn P, int64_t n, const uint8_t* __restrict__ st, const int32_t* __restrict__ slot_of,
    const uint8_t* __restrict__ cR, const gf_seqmatch* __restrict__ mR, uint32_t* __restrict__ tile_hc,
    uint32_t* __restrict__ tile_hb) {
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  const int64_t p0 = (int64_t)blockIdx.x * GF_PTILE + (int64_t)threadIdx.x * GF_PPER;
  int hc = 0;
  long long hb = 0;
  for (int k = 0; k < GF_PPER; ++k) {
    const int64_t p = p0 + k;
    if (p >= n) break;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (st[3 * p + s] == GF_PS_NONE) continue;
      int slot;
      if (gf_pair_final(P, p, s, st, slot_of, cR, mR, slot)) {
        const uint8_t* b; const uint8_t* q; int32_t len; uint8_t cnt; const gf_seqmatch* m;
        gf_pair_candidate(P, p, s, b, q, len, cnt, m);
        hc += 1;
        hb += len;
      }
    }
  }
  int ea, ta; long long eb, tb;
  gf_block_scan2(hc, hb, s_a, s_b, ea, eb, ta, tb);
  if (threadIdx.x == 0) {
    tile_hc[blockIdx.x] = (uint32_t)ta;
    tile_hb[blockIdx.x] = (uint32_t)tb;
  }
}

__global__ __launch_bounds__(GF_CTHREADS) void gf_k_pair_final_write(
    GfPairIn P, int64_t n, int64_t pair_id_base, const uint8_t* __restrict__ st, const int32_t* __restrict__ slot_of,
    const uint8_t* __restrict__ cR, const gf_seqmatch* __restrict__ mR, const int64_t* __restrict__ r_off,
    const uint8_t* __restrict__ r_bases, const uint8_t* __restrict__ r_quals, const int64_t* __restrict__ tile_off_hc,
    const int64_t* __restrict__ tile_off_hb, gf_pair_hit* __restrict__ hits, int64_t hits_cap,
    uint8_t* __restrict__ out_bases, uint8_t* __restrict__ out_quals, int64_t bytes_cap) {
  __shared__ int s_a[4];
  __shared__ long long s_b[4];
  const int64_t p0 = (int64_t)blockIdx.x * GF_PTILE + (int64_t)threadIdx.x * GF_PPER;
  int hc = 0;
  long long hb = 0;
  for (int k = 0; k < GF_PPER; ++k) {
    const int64_t p = p0 + k;
    if (p >= n) break;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (st[3 * p + s] == GF_PS_NONE) continue;
      int slot;
      if (gf_pair_final(P, p, s, st, slot_of, cR, mR, slot)) {
        const uint8_t* b; const uint8_t* q; int32_t len; uint8_t cnt; const gf_seqmatch* m;
        gf_pair_candidate(P, p, s, b, q, len, cnt, m);
        hc += 1;
        hb += len;
      }
    }
  }
  int ea, ta; long long eb, tb;
  gf_block_scan2(hc, hb, s_a, s_b, ea, eb, ta, tb);
  if (__ballot(hc != 0) == 0) return;  // (whole wavefronts: the reads are written by all 64 lanes)
  int64_t k_out = tile_off_hc[blockIdx.x] + ea;
  int64_t b_out = tile_off_hb[blockIdx.x] + eb;
  const int64_t p = p0;  // GF_PPER == 1
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    int slot = -1;
    const bool mine = p < n && st[3 * p + s] != GF_PS_NONE && gf_pair_final(P, p, s, st, slot_of, cR, mR, slot);
    const uint8_t* b = nullptr; const uint8_t* q = nullptr; int32_t len = 0; uint8_t cnt; const gf_seqmatch* m;
    bool bytes_fit = false;
    GfPairQual Q = gf_pair_qual_none();
    if (mine) {
      gf_pair_candidate(P, p, s, b, q, len, cnt, m);
      if (slot >= 0) {  // the match is on the reverse complement: its bases, its mapping
        b = r_bases + r_off[slot];
        q = r_quals + r_off[slot];
        m = mR + 2 * (int64_t)slot;
      }
      if (k_out < hits_cap) {
        gf_pair_hit h;
        h.pair_id = pair_id_base + p;
        h.source = s;
        // bit 0: found on the reverse complement; bit 1: ReadMatch.m_reversed as the reference sets it —
        // for R1 / R2 (pescanner.rs:489,:511), not for a merged read (:465-468)
        h.flags = (slot >= 0 ? 1 : 0) | ((slot >= 0 && s != 0) ? 2 : 0);
        h.read_len = len;
        h.merge_diff = s == 0 ? P.m_diff[p] : 0;
        h.seq_offset = b_out;
        h.m[0] = m[0];
        h.m[1] = m[1];
        hits[k_out] = h;
      }
      bytes_fit = b_out + len <= bytes_cap;
      if (bytes_fit) Q = gf_pair_qual(P, p, s, q);  // (q: the retry batch's for a reverse complement)
    }
    gf_wave_write_reads(__ballot(bytes_fit), b, Q, len, (long long)b_out, out_bases, out_quals, false);
    if (mine) {
      k_out += 1;
      b_out += len;
    }
  }
}

__global__ void gf_k_pair_totals(const int64_t* __restrict__ d_hits, const int64_t* __restrict__ d_hit_bytes,
                                 const int64_t* __restrict__ n_merged, int64_t hits_cap, int64_t bytes_cap,
                                 int64_t* __restrict__ totals) {
  totals[0] = *d_hits;
  totals[1] = *d_hit_bytes;
  totals[2] = *n_merged;
  if (*d_hits > hits_cap || *d_hit_bytes > bytes_cap) totals[4] |= 2;
}
