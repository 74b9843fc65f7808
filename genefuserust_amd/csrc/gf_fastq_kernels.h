// SURVEY.md §8(f)-2 — FASTQ ingest on the device: FastqReader::read
// (src/core/fastq_reader.rs:75-147) for a whole text buffer resident in HBM.
//
// The reference reads four lines per record with BufRead::read_line (name, sequence,
// strand, quality), strips one trailing '\n' from each (a '\r' stays part of the line),
// accepts a last line without '\n', and stops at the first record that has fewer than four
// lines.  On the device that is: find every '\n' (ordered positions: count per tile, scan
// the tile totals, write), then record i owns lines 4i .. 4i+3, and its sequence and quality
// lines are copied into the back-to-back `bases` / `quals` + `offsets` layout that
// gf_map_reads_device and gf_fast_merge_*_device take.  (gzip stays on the host.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf_compact_kernels.h"

#define GF_FQ_TILE 16384                        // text bytes per block
#define GF_FQ_PER (GF_FQ_TILE / GF_CTHREADS)    // 64 contiguous bytes per thread

// bit 7 of each byte of the result = that byte of x is '\n'
__device__ __forceinline__ uint32_t gf_newline_flags(uint32_t x) {
  const uint32_t v = x ^ 0x0A0A0A0Au;
  return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}

// the thread's 64 bytes as 16 dwords; bytes at or beyond n read as 0
__device__ __forceinline__ void gf_fq_load64(const uint8_t* __restrict__ text, int64_t n, int64_t p0, uint32_t (&w)[16]) {
  if (p0 + GF_FQ_PER <= n && (((uintptr_t)(text + p0)) & 15u) == 0) {
    const uint4* q = (const uint4*)(text + p0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint4 v = q[k];
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint32_t x = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t p = p0 + 4 * k + b;
        if (p < n) x |= (uint32_t)text[p] << (8 * b);
      }
      w[k] = x;
    }
  }
}

// one bit per byte of the thread's 64: bit 4k+b = byte b of dword k is '\n'
__device__ __forceinline__ uint64_t gf_fq_mask64(const uint32_t (&w)[16]) {
  uint64_t m = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t f = gf_newline_flags(w[k]);  // bits 7, 15, 23, 31
    m |= (uint64_t)((((f >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);
  }
  return m;
}

// per tile: the number of newlines; per thread: the newline mask of its 64 bytes, kept for
// gf_k_fq_write (an eighth of the text to read back instead of the text itself)
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_fq_count(const uint8_t* __restrict__ text, int64_t n,
                                                             uint32_t* __restrict__ tile_counts,
                                                             uint64_t* __restrict__ masks) {
  __shared__ int s_wave[4];
  const int64_t p0 = (int64_t)blockIdx.x * GF_FQ_TILE + (int64_t)threadIdx.x * GF_FQ_PER;
  uint64_t m = 0;
  if (p0 < n) {
    uint32_t w[16];
    gf_fq_load64(text, n, p0, w);
    m = gf_fq_mask64(w);
  }
  masks[(int64_t)blockIdx.x * GF_CTHREADS + threadIdx.x] = m;
  int total;
  gf_block_exclusive_scan(__popcll(m), s_wave, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = (uint32_t)total;
}

// nl_pos[k] = byte offset of the k-th '\n' (k < cap); *n_lines = lines in the text, counting
// a last line without '\n'
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_fq_write(const uint8_t* __restrict__ text, int64_t n,
                                                             const int64_t* __restrict__ tile_offsets,
                                                             const int64_t* __restrict__ n_newlines,
                                                             const uint64_t* __restrict__ masks,
                                                             int64_t* __restrict__ nl_pos, int64_t cap,
                                                             int64_t* __restrict__ n_lines) {
  __shared__ int s_wave[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_lines = *n_newlines + ((n > 0 && text[n - 1] != '\n') ? 1 : 0);
  const int64_t p0 = (int64_t)blockIdx.x * GF_FQ_TILE + (int64_t)threadIdx.x * GF_FQ_PER;
  uint64_t m = masks[(int64_t)blockIdx.x * GF_CTHREADS + threadIdx.x];
  const int c = __popcll(m);
  int total;
  int64_t pos = tile_offsets[blockIdx.x] + gf_block_exclusive_scan(c, s_wave, &total);
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    if (pos < cap) nl_pos[pos] = p0 + b;
    ++pos;
  }
}

// [start, end) of line k (k < n_lines)
__device__ __forceinline__ void gf_fq_line(const int64_t* __restrict__ nl_pos, int64_t n_newlines, int64_t n_bytes,
                                           int64_t k, int64_t& start, int64_t& end) {
  start = k ? nl_pos[k - 1] + 1 : 0;
  end = k < n_newlines ? nl_pos[k] : n_bytes;
}

// One wavefront copies n bytes from src to dst (any alignments): whole destination dwords
// are assembled from two aligned source dwords with v_alignbyte, the few bytes before the
// first and after the last whole dword are copied singly.  An aligned dword never crosses a
// page, and every dword loaded holds at least one byte of the line, so nothing outside the
// text's pages is touched.
__device__ __forceinline__ void gf_fq_copy_line(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int n,
                                                int lane) {
  const int head = (int)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u);  // bytes before the first aligned dword
  const int h = head < n ? head : n;
  const int ndw = (n - h) >> 2;
  const int tail0 = h + 4 * ndw;
  if (lane < h) dst[lane] = src[lane];
  if (lane >= 32 && lane - 32 < n - tail0) dst[tail0 + lane - 32] = src[tail0 + lane - 32];
  uint32_t* d4 = (uint32_t*)(dst + h);
  const uint8_t* s0 = src + h;
  const uint32_t sh = (uint32_t)((uintptr_t)s0 & 3u);
  const uint32_t* s4 = (const uint32_t*)(s0 - sh);
  for (int k = lane; k < ndw; k += 64) {
    const uint32_t lo = s4[k];
    const uint32_t hi = sh ? s4[k + 1] : 0u;
    d4[k] = __builtin_amdgcn_alignbyte(hi, lo, sh);
  }
}

#define GF_FQ_RTILE 256  // records per block of the two record kernels: one per thread

// sequence length of every record, summed per tile of 256 records
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_fq_lens(const int64_t* __restrict__ nl_pos, int64_t n_newlines,
                                                            int64_t n_bytes, int64_t n_rec,
                                                            uint32_t* __restrict__ tile_counts) {
  __shared__ int s_wave[4];
  const int64_t r = (int64_t)blockIdx.x * GF_FQ_RTILE + threadIdx.x;
  int c = 0;
  if (r < n_rec) {
    int64_t s, e;
    gf_fq_line(nl_pos, n_newlines, n_bytes, 4 * r + 1, s, e);
    c = (int)(e - s);
  }
  int total;
  gf_block_exclusive_scan(c, s_wave, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = (uint32_t)total;
}

// 16 bytes from any address (the hardware takes unaligned global loads)
struct __attribute__((packed, aligned(1))) GfBytes16 { uint32_t v[4]; };
__device__ __forceinline__ uint4 gf_fq_load16(const uint8_t* p) {
  const GfBytes16 t = *(const GfBytes16*)p;
  return make_uint4(t.v[0], t.v[1], t.v[2], t.v[3]);
}
// bytes 0 .. k-1 of a, bytes k .. 15 of b
__device__ __forceinline__ uint4 gf_fq_splice16(uint4 a, uint4 b, int k) {
  uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = k - 4 * j;  // bytes of this dword that come from a
    const uint32_t m = n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << (8 * n)) - 1u));
    o[j] = (av[j] & m) | (bv[j] & ~m);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// offsets[] of the records, then the sequence and quality lines of a tile of 256 records
// copied as ONE ragged memcpy per output array: the tile's records are contiguous in `bases`
// and `quals`, so a thread takes an aligned 16-byte piece of the output, finds the record it
// falls in (binary search over the tile's offsets in LDS) and fetches its 16 bytes from the
// text with one unaligned load — two loads spliced when the piece spans the end of one record
// and the start of the next; the pieces at the tile's edges (shared with the neighbouring
// blocks) and pieces over more than two records go byte by byte.  Every lane moves 32 bytes
// per round whatever the read length (a wavefront per record kept 38 of 64 lanes busy with
// 4 bytes each on 150-base reads).  A quality line shorter than its sequence is padded with
// '!' (Phred 0), a longer one is cut: both are counted in *n_bad (the reference does not
// check, and its fast_merge would panic on the short ones); a tile that holds such a record,
// or that does not fit the caller's buffers, is copied a wavefront per record instead.
// WITH_Q = false (r03 b): the qualities stay where they are — 1.5 GB per 10 M records that a few thousand hit records
// and the merge's rare mismatching columns ever read — and `qual_off[r]` says where record r's quality line starts in
// the text; a record whose quality line has another length than its sequence is counted in *n_bad as ever (the
// caller then gathers with qualities: the in-place form has no room for the padding).
template <bool WITH_Q>
__global__ __launch_bounds__(GF_CTHREADS) void gf_k_fq_gather(const uint8_t* __restrict__ text,
                                                              const int64_t* __restrict__ nl_pos, int64_t n_newlines,
                                                              int64_t n_bytes, int64_t n_rec,
                                                              const int64_t* __restrict__ tile_offsets,
                                                              int64_t* __restrict__ offsets,
                                                              uint8_t* __restrict__ bases, uint8_t* __restrict__ quals,
                                                              int64_t cap_bytes, unsigned long long* __restrict__ n_bad,
                                                              int64_t* __restrict__ qual_off) {
  __shared__ int s_wave[4];
  __shared__ int64_t s_ss[GF_FQ_RTILE], s_qs[GF_FQ_RTILE];
  __shared__ int s_rel[GF_FQ_RTILE + 2], s_qlen[GF_FQ_RTILE];
  __shared__ int s_odd;
  if (threadIdx.x == 0) s_odd = 0;
  const int64_t t0 = (int64_t)blockIdx.x * GF_FQ_RTILE;
  const int64_t r = t0 + threadIdx.x;
  int len = 0, qlen = 0;
  int64_t ss = 0, qs = 0;
  if (r < n_rec) {
    int64_t se, qe;
    gf_fq_line(nl_pos, n_newlines, n_bytes, 4 * r + 1, ss, se);
    gf_fq_line(nl_pos, n_newlines, n_bytes, 4 * r + 3, qs, qe);
    len = (int)(se - ss);
    qlen = (int)(qe - qs);
  }
  int total;
  const int rel = gf_block_exclusive_scan(len, s_wave, &total);  // (its barriers order the s_odd store above)
  const int64_t pos0 = tile_offsets[blockIdx.x];
  if (r < n_rec) {
    offsets[r] = pos0 + rel;
    if (!WITH_Q) qual_off[r] = qs;
    if (r == n_rec - 1) offsets[n_rec] = pos0 + rel + len;
    if (qlen != len) {
      atomicAdd(n_bad, 1ull);
      if (WITH_Q) s_odd = 1;  // (the careful path is about the qualities' padding)
    }
  }
  s_ss[threadIdx.x] = ss; s_qs[threadIdx.x] = qs;
  s_rel[threadIdx.x] = rel;  // (threads past the last record hold `total`)
  s_qlen[threadIdx.x] = qlen;
  if (threadIdx.x < 2) s_rel[GF_FQ_RTILE + threadIdx.x] = total;
  __syncthreads();
  const int64_t end = pos0 + total;
  const bool aligned = (((uintptr_t)bases | (WITH_Q ? (uintptr_t)quals : (uintptr_t)0)) & 15u) == 0;
  if (!s_odd && end <= cap_bytes && aligned) {
    const int64_t c_hi = (end + 15) >> 4;
    for (int64_t c = (pos0 >> 4) + threadIdx.x; c < c_hi; c += GF_CTHREADS) {
      const int64_t b0 = c << 4;
      const int64_t lo = b0 > pos0 ? b0 : pos0, hi = b0 + 16 < end ? b0 + 16 : end;  // this tile's bytes of the piece
      if (lo >= hi) continue;
      // the record that holds byte lo: the largest i with s_rel[i] <= x (then s_rel[i + 1] > x)
      const int x = (int)(lo - pos0);
      int i = 0, j = GF_FQ_RTILE;
      while (j - i > 1) {
        const int mid = (i + j) >> 1;
        if (s_rel[mid] <= x) i = mid; else j = mid;
      }
      const int xb = (int)(b0 - pos0);  // (negative in the tile's first piece when it starts mid-piece)
      const bool whole = lo == b0 && hi == b0 + 16;
      const int64_t sa = s_ss[i] + (xb - s_rel[i]), qa = s_qs[i] + (xb - s_rel[i]);
      if (whole && s_rel[i + 1] >= xb + 16) {
        *(uint4*)(bases + b0) = gf_fq_load16(text + sa);
        if (WITH_Q) *(uint4*)(quals + b0) = gf_fq_load16(text + qa);
      } else if (whole && s_rel[i + 2] >= xb + 16 &&
                 (WITH_Q ? qa + 16 <= n_bytes : (sa + 16 <= n_bytes && s_ss[i + 1] + 16 <= n_bytes))) {
        // (sa < qa: the sequence line's load cannot run off the text when the quality line's does not;
        //  the second record's loads start k bytes before its lines, inside the text)
        const int k = s_rel[i + 1] - xb;  // bytes of record i in this piece, 1 .. 15
        *(uint4*)(bases + b0) = gf_fq_splice16(gf_fq_load16(text + sa), gf_fq_load16(text + s_ss[i + 1] - k), k);
        if (WITH_Q) *(uint4*)(quals + b0) = gf_fq_splice16(gf_fq_load16(text + qa), gf_fq_load16(text + s_qs[i + 1] - k), k);
      } else {
        for (int64_t b = lo; b < hi; ++b) {
          const int xx = (int)(b - pos0);
          while (s_rel[i + 1] <= xx) ++i;
          bases[b] = text[s_ss[i] + (xx - s_rel[i])];
          if (WITH_Q) quals[b] = text[s_qs[i] + (xx - s_rel[i])];
        }
      }
    }
    return;
  }
  // the careful path: a wavefront per record
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int in_tile = (int)((n_rec - t0) < GF_FQ_RTILE ? (n_rec - t0) : GF_FQ_RTILE);
  for (int i = wave; i < in_tile; i += GF_CTHREADS / 64) {
    const int64_t dst = pos0 + s_rel[i];
    const int ln = s_rel[i + 1] - s_rel[i], ql = s_qlen[i];
    if (dst + ln > cap_bytes) continue;  // (the caller's buffers are too small: offsets still tell how much is needed)
    gf_fq_copy_line(bases + dst, text + s_ss[i], ln, lane);
    if (WITH_Q) {
      const uint8_t* qsrc = text + s_qs[i];
      for (int k = lane; k < ln; k += 64) quals[dst + k] = k < ql ? qsrc[k] : (uint8_t)'!';
    }
  }
}
