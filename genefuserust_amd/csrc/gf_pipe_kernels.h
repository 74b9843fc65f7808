// Flat ("thread per read") form of Indexer::map_read's first pass for reads of up
// to 256 bases — same exact decisions as gf_map_kernels.h, far fewer instructions.
//
// The wave-per-read kernel spends ~640 wave-instructions per read, most of them
// wave-uniform bookkeeping (ballots, scalar branches, LDS staging) that serve one
// read at a time; at 1.6 G reads/s the scalar/vector issue ports, not memory, are
// the limit.  Here every lane owns a read, so the same bookkeeping is ordinary
// per-lane arithmetic shared by 64 reads per instruction:
//
//   K_pack        thread per 16 bases: ASCII -> per-read record in HBM
//                 (2 bits per base + an "unusable base" stream in the same layout)
//   K_seedverify  thread per read: 4 seed probes; each UNIQUE seed hit names a
//                 candidate diagonal K; K is verified against both strands of the
//                 genes laid out in site-code space (gf_table.h: gd, ub) with
//                 word-parallel bit tricks (16 bases per XOR): window i counts for K
//                 iff its 16 bases equal the bases of site K+i and that site is the
//                 only site of its key.  Then the exact bound of gf_map_kernels.h
//                 ("a diagonal gets at most one vote per window that can still vote"):
//                   v1 + open < 20 or v2 + open < 10  ->  []   (decided, nothing probed)
//                 otherwise the read goes to K_probe with (v1, v2, verified mask).
//   K_probe       thread per undecided read: probes its unverified windows one by one
//                 (one 64-byte bucket per probe), h = windows that voted; stops as soon
//                 as v1 + h + left < 20 or v2 + h + left < 10 -> [].  Reads that
//                 survive (junction reads, repeats) go to the list for
//   K_full        the wave-per-read kernel (gf_k_map_reads_list), which recomputes the
//                 read from scratch — votes, top two, gate, second pass, segments.
// Every read that ends here with [] was *proved* to fail the gate of
// indexer.rs:353-360; everything else is computed by the exact kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"
#include "gf_map_kernels.h"
#include "gf_table.h"

// words of 16 bases per read record for reads up to LMAX bases (+1 so that a window
// starting in the last word can always read the following word)
#define GF_PW(LMAX) (((LMAX) + 15) / 16 + 1)
// record = pk[PW] | iv2[PW], padded to a multiple of 4 words (16-byte vector loads)
#define GF_RW(PW) ((2 * (PW) + 3) / 4 * 4)

struct GfPipeEntry {  // one undecided read handed from K_seedverify to K_probe (32 B)
  uint32_t read;      // read index in the batch
  uint32_t v1v2;      // v1 | v2 << 8
  uint32_t todo[4];   // bit w = stride-2 window w is clean and not verified: probe it
  uint32_t pad[2];
};

// 4 ASCII bases (little-endian dword) -> codes in 2-bit layout + "valid" bits in the
// same layout (bit 2k = base k is one of A,C,G,T; indexer.rs:825-841)
__device__ __forceinline__ void gf_convert4_2bit(uint32_t x, uint32_t& code8, uint32_t& val8) {
  uint32_t y = (x >> 1) & 0x03030303u;
  code8 = (y * 0x01041040u) >> 24;
  uint32_t ok = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t b = (x >> (8 * j)) & 0xFFu;
    uint32_t v = (((b & 0xE0u) == 0x40u) ? 1u : 0u) & (0x0010008Au >> (b & 31u));
    ok |= v << (8 * j);
  }
  val8 = (ok * 0x01041040u) >> 24;  // bit 0 of byte k -> bit 2k
}

// ---- K_pack: thread per (read, word) ----
template <int PW>
__global__ __launch_bounds__(256) void gf_k_pack(const uint8_t* __restrict__ bases,
                                                 const int64_t* __restrict__ offsets, int64_t n, int lmin,
                                                 int lmax, uint32_t* __restrict__ rec) {
  constexpr int RW = GF_RW(PW);
  const int64_t total = n * PW;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / PW;
    const int j = (int)(t - r * PW);
    const int64_t off0 = offsets[r];
    const int64_t len64 = offsets[r + 1] - off0;
    if (len64 <= lmin || len64 > lmax) continue;  // another length class owns this read
    const int L = (int)len64;
    const int b0 = 16 * j;  // first base of this word
    uint32_t pk = 0, iv = 0x55555555u;
    if (b0 < L) {
      const uintptr_t addr = (uintptr_t)(bases + off0 + b0);
      const uint32_t sh = (uint32_t)(addr & 3u);
      const uint32_t* pw = (const uint32_t*)(addr - sh);
      const int nbytes = (L - b0) < 16 ? (L - b0) : 16;
      const int ndw = (int)((sh + (uint32_t)nbytes + 3u) >> 2);  // aligned dwords that overlap the read
      uint32_t d[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) d[k] = (k < ndw) ? pw[k] : 0u;
      uint32_t val = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t x = sh ? __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh) : d[k];
        uint32_t c8, v8;
        gf_convert4_2bit(x, c8, v8);
        pk |= c8 << (8 * k);
        val |= v8 << (8 * k);
      }
      // bases at or beyond the end of the read are unusable
      const uint32_t inside = nbytes >= 16 ? 0x55555555u : ((1u << (2 * nbytes)) - 1u) & 0x55555555u;
      iv = (val & inside) ^ 0x55555555u;
    }
    rec[r * RW + j] = pk;
    rec[r * RW + PW + j] = iv;
  }
}

// 32 bits starting at bit position 2*p of a little-endian stream held in an unrolled
// register array (p compile-time after unrolling)
#define GF_STREAM_AT(arr, p) gf_window((arr)[(p) >> 4], (arr)[((p) >> 4) + 1], (uint32_t)(p))

// For a stream z (bit 2p set = base p is bad), bit 2p of the result word j is set iff
// bases p .. p+15 are all good ("a clean window starts at p").  PW-1 result words.
template <int PW>
__device__ __forceinline__ void gf_clean_windows(const uint32_t (&z)[PW], uint32_t (&out)[PW]) {
  uint32_t g[PW + 1];
#pragma unroll
  for (int j = 0; j < PW; ++j) g[j] = ~z[j] & 0x55555555u;
  g[PW] = 0;
  // runs of 2, 4, 8, 16 good bases by doubling (funnel shifts across words)
#pragma unroll
  for (int s = 1; s <= 8; s <<= 1) {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const uint32_t sh = 2u * s;  // bits
      const uint32_t nxt = sh == 32u ? g[j + 1] : ((g[j] >> sh) | (g[j + 1] << (32u - sh)));
      g[j] &= nxt;
    }
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) out[j] = g[j];
}

// every 4th bit (bits 0,4,..,28) of x gathered into the low 8 bits
__device__ __forceinline__ uint32_t gf_gather_nibble_lsb(uint32_t x) {
  x &= 0x11111111u;
  x = (x | (x >> 3)) & 0x03030303u;
  x = (x | (x >> 6)) & 0x000F000Fu;
  x = (x | (x >> 12)) & 0xFFu;
  return x;
}

// first bucket of a lookup with the loads issued by the caller (ILP over several keys)
__device__ __forceinline__ uint32_t gf_match_bucket(uint4 q0, uint4 q1, uint4 q2, uint4 q3, uint32_t key,
                                                    bool& overflow) {
  uint32_t r = 0;
  r = (q0.y == key && (q0.x & GF_VAL_LOW)) ? q0.x : r;
  r = (q0.w == key && (q0.z & GF_VAL_LOW)) ? q0.z : r;
  r = (q1.y == key && (q1.x & GF_VAL_LOW)) ? q1.x : r;
  r = (q1.w == key && (q1.z & GF_VAL_LOW)) ? q1.z : r;
  r = (q2.y == key && (q2.x & GF_VAL_LOW)) ? q2.x : r;
  r = (q2.w == key && (q2.z & GF_VAL_LOW)) ? q2.z : r;
  r = (q3.y == key && (q3.x & GF_VAL_LOW)) ? q3.x : r;
  r = (q3.w == key && (q3.z & GF_VAL_LOW)) ? q3.z : r;
  overflow = !r && (q0.x & GF_VAL_OVF);
  return r & GF_VAL_LOW;
}

// ---- K_seedverify: thread per read ----
template <int PW>
__global__ __launch_bounds__(256) void gf_k_seedverify(GfTable T, const int64_t* __restrict__ offsets, int64_t n,
                                                       int lmin, int lmax, int mark_too_long,
                                                       const uint32_t* __restrict__ rec,
                                                       uint8_t* __restrict__ counts,
                                                       GfPipeEntry* __restrict__ list_b,
                                                       unsigned int* __restrict__ n_b) {
  constexpr int RW = GF_RW(PW);
  constexpr int NSEED = 4;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t len64 = offsets[r + 1] - offsets[r];
    bool undecided = false;
    uint32_t e_v1v2 = 0, e_mask[4] = {0, 0, 0, 0};
    if (len64 > lmax) {
      if (mark_too_long) counts[r] = GF_COUNT_TOO_LONG;
    } else if (len64 <= lmin) {
      // another length class owns this read
    } else if (len64 < GF_KMER + 2 * (GF_MAJOR_KEYS / 2 - 1)) {
      counts[r] = 0;  // fewer than 20 stride-2 windows: count1 < 20 whatever they hit
    } else {
      // the read's record: codes and unusable-base stream
      uint32_t pk[PW], iv[PW];
      const uint4* rp = (const uint4*)(rec + r * RW);
      uint32_t tmp[RW];
#pragma unroll
      for (int q = 0; q < RW / 4; ++q) {
        const uint4 v = rp[q];
        tmp[4 * q] = v.x; tmp[4 * q + 1] = v.y; tmp[4 * q + 2] = v.z; tmp[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < PW; ++j) { pk[j] = tmp[j]; iv[j] = tmp[PW + j]; }

      // clean stride-2 windows of the read (all 16 bases usable): bit 4w' of word w/8
      uint32_t cw[PW];
      gf_clean_windows<PW>(iv, cw);
      int nvalid = 0;
#pragma unroll
      for (int j = 0; j < PW - 1; ++j) nvalid += __popc(cw[j] & 0x11111111u);

      // seeds at bases 0, 32, 64, 96 (word aligned: key = one record word), two
      // probes in flight at a time
      uint32_t cand[NSEED];
#pragma unroll
      for (int s0 = 0; s0 < NSEED; s0 += 2) {
        uint4 q[2][4];
        bool ok[2];
        uint32_t key[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int s = s0 + u;
          const int wj = 2 * s < PW - 1 ? 2 * s : 0;
          ok[u] = (2 * s < PW - 1) && (cw[wj] & 1u);
          key[u] = pk[wj];
          const uint32_t b = gf_bucket_of(key[u], T.nbuckets);
          const uint4* p = (const uint4*)(T.slots + (size_t)b * GF_SLOTS_PER_BUCKET);
          if (ok[u]) { q[u][0] = p[0]; q[u][1] = p[1]; q[u][2] = p[2]; q[u][3] = p[3]; }
          else { q[u][0] = q[u][1] = q[u][2] = q[u][3] = make_uint4(0, 0, 0, 0); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int s = s0 + u;
          uint32_t val = 0;
          if (ok[u]) {
            bool ovf;
            val = gf_match_bucket(q[u][0], q[u][1], q[u][2], q[u][3], key[u], ovf);
            if (ovf) val = gf_lookup(T, key[u]);  // rare: the key may live in a later bucket
          }
          cand[s] = ((val >> GF_TYPE_SHIFT) == GF_TYPE_UNIQUE) ? (val & GF_LIN_MASK) - 32u * s : GF_NONE_LIN;
        }
      }

      // verify each distinct candidate diagonal
      int v1 = 0, v2 = 0, nver = 0;
      uint32_t vm[PW];  // verified windows, same sparse layout as cw
#pragma unroll
      for (int j = 0; j < PW; ++j) vm[j] = 0;
#pragma unroll
      for (int s = 0; s < NSEED; ++s) {
        bool fresh = cand[s] != GF_NONE_LIN;
#pragma unroll
        for (int s2 = 0; s2 < s; ++s2) fresh = fresh && cand[s2] != cand[s];
        if (fresh) {
          const uint32_t K = cand[s];
          const uint32_t w0 = K >> 4, bo = 2u * (K & 15u);
          uint32_t gdr[PW + 1], ubr[PW + 1];
#pragma unroll
          for (int j = 0; j < PW + 1; ++j) { gdr[j] = T.gd[w0 + j]; ubr[j] = T.ub2[w0 + j]; }
          int cnt = 0;
#pragma unroll
          for (int j = 0; j < PW - 1; ++j) {  // the last record word only feeds the previous one
            const uint32_t g = bo ? ((gdr[j] >> bo) | (gdr[j + 1] << (32u - bo))) : gdr[j];
            const uint32_t x = pk[j] ^ g;
            const uint32_t bad = ((x | (x >> 1)) & 0x55555555u) | iv[j];
            gdr[j] = bad;  // reuse as the "bad base" stream of this candidate
          }
          {
            const int j = PW - 1;
            const uint32_t g = bo ? ((gdr[j] >> bo) | (gdr[j + 1] << (32u - bo))) : gdr[j];
            const uint32_t x = pk[j] ^ g;
            gdr[j] = ((x | (x >> 1)) & 0x55555555u) | iv[j];
          }
          uint32_t zz[PW], cl[PW];
#pragma unroll
          for (int j = 0; j < PW; ++j) zz[j] = gdr[j];
          gf_clean_windows<PW>(zz, cl);
#pragma unroll
          for (int j = 0; j < PW - 1; ++j) {
            const uint32_t u = bo ? ((ubr[j] >> bo) | (ubr[j + 1] << (32u - bo))) : ubr[j];
            const uint32_t ver = cl[j] & u & 0x11111111u & ~vm[j];
            vm[j] |= ver;
            cnt += __popc(ver);
          }
          nver += cnt;
          if (cnt > v1) { v2 = v1; v1 = cnt; } else if (cnt > v2) { v2 = cnt; }
        }
      }

      // every other diagonal gets at most one vote per window that can still vote
      const int open = nvalid - nver;
      if (v1 + open < GF_MAJOR_KEYS / 2 || v2 + open < GF_MINOR_KEYS / 2) {
        counts[r] = 0;
      } else {
        undecided = true;
        e_v1v2 = (uint32_t)v1 | ((uint32_t)v2 << 8);
#pragma unroll
        for (int j = 0; j < PW - 1; ++j) {
          const uint32_t byte = gf_gather_nibble_lsb(cw[j] & ~vm[j]);
          e_mask[j >> 2] |= byte << (8 * (j & 3));
        }
      }
    }
    // wave-aggregated append to the undecided list
    const uint64_t m = __ballot(undecided);
    if (m) {
      unsigned int base = 0;
      const int leader = __builtin_ctzll(m);
      const int lane = threadIdx.x & 63;
      if (lane == leader) base = atomicAdd(n_b, (unsigned int)__popcll(m));
      base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
      if (undecided) {
        GfPipeEntry e;
        e.read = (uint32_t)r;
        e.v1v2 = e_v1v2;
        e.todo[0] = e_mask[0]; e.todo[1] = e_mask[1]; e.todo[2] = e_mask[2]; e.todo[3] = e_mask[3];
        e.pad[0] = e.pad[1] = 0;
        list_b[base + gf_lanes_below(m)] = e;
      }
    }
  }
}

// ---- K_probe: thread per undecided read ----
template <int PW>
__global__ __launch_bounds__(256) void gf_k_probe(GfTable T, const uint32_t* __restrict__ rec,
                                                  const GfPipeEntry* __restrict__ list_b,
                                                  const unsigned int* __restrict__ n_b,
                                                  uint8_t* __restrict__ counts, uint32_t* __restrict__ list_c,
                                                  unsigned int* __restrict__ n_c) {
  constexpr int RW = GF_RW(PW);
  // the read's codes live in LDS for the duration of its probes ([word][thread]: each
  // thread reads only its own column, conflict-free), not in re-fetched HBM lines
  __shared__ uint32_t s_pk[PW * 256];
  const unsigned int nb = *n_b;
  const unsigned int nb_round = (nb + 63u) & ~63u;  // whole waves stay in the loop for the ballot
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < nb_round; t += gridDim.x * blockDim.x) {
    bool to_full = false;
    uint32_t r = 0;
    if (t < nb) {
      const GfPipeEntry e = list_b[t];
      r = e.read;
      const int v1 = (int)(e.v1v2 & 0xFFu), v2 = (int)((e.v1v2 >> 8) & 0xFFu);
      uint32_t m0 = e.todo[0], m1 = e.todo[1], m2 = e.todo[2], m3 = e.todo[3];
      int left = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
      int h = 0;
      {
        const uint4* rp = (const uint4*)(rec + (size_t)r * RW);
#pragma unroll
        for (int q = 0; q < (PW + 3) / 4; ++q) {
          const uint4 v = rp[q];
          s_pk[(4 * q) * 256 + threadIdx.x] = v.x;
          if (4 * q + 1 < PW) s_pk[(4 * q + 1) * 256 + threadIdx.x] = v.y;
          if (4 * q + 2 < PW) s_pk[(4 * q + 2) * 256 + threadIdx.x] = v.z;
          if (4 * q + 3 < PW) s_pk[(4 * q + 3) * 256 + threadIdx.x] = v.w;
        }
      }
      // count1 <= v1 + h + left and count2 <= v2 + h + left (one vote per window per diagonal)
      bool dead = (v1 + left < GF_MAJOR_KEYS / 2) || (v2 + left < GF_MINOR_KEYS / 2);
      while (!dead && left > 0) {
        int w;
        if (m0) { w = __builtin_ctz(m0); m0 &= m0 - 1; }
        else if (m1) { w = 32 + __builtin_ctz(m1); m1 &= m1 - 1; }
        else if (m2) { w = 64 + __builtin_ctz(m2); m2 &= m2 - 1; }
        else { w = 96 + __builtin_ctz(m3); m3 &= m3 - 1; }
        const int j = w >> 3;
        const uint32_t sh = 4u * (uint32_t)(w & 7);
        const uint32_t lo = s_pk[j * 256 + threadIdx.x], hi = s_pk[(j + 1) * 256 + threadIdx.x];
        const uint32_t key = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
        uint32_t ty = 0;
        bool maybe = true;
        if (T.bloom_words) {
          const uint32_t h2 = GF_BLOOM_H2(gf_mix32(key));
          const uint32_t bits = GF_BLOOM_BITS(h2);
          maybe = (T.bloom[GF_BLOOM_WORD(h2, T.bloom_words)] & bits) == bits;
        }
        if (maybe) ty = gf_lookup(T, key) >> GF_TYPE_SHIFT;
        h += (ty == GF_TYPE_UNIQUE || ty == GF_TYPE_DUPES) ? 1 : 0;
        left -= 1;
        dead = (v1 + h + left < GF_MAJOR_KEYS / 2) || (v2 + h + left < GF_MINOR_KEYS / 2);
      }
      if (dead) counts[r] = 0;
      else to_full = true;
    }
    const uint64_t m = __ballot(to_full);
    if (m) {
      unsigned int base = 0;
      const int leader = __builtin_ctzll(m);
      const int lane = threadIdx.x & 63;
      if (lane == leader) base = atomicAdd(n_c, (unsigned int)__popcll(m));
      base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
      if (to_full) list_c[base + gf_lanes_below(m)] = r;
    }
  }
}

// ---- K_full: the exact wave-per-read kernel over a list of read indices ----
template <int LCAP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gf_k_map_reads_list(GfTable T, const uint8_t* __restrict__ bases,
                                                                  const int64_t* __restrict__ offsets,
                                                                  const uint32_t* __restrict__ list,
                                                                  const unsigned int* __restrict__ n_list,
                                                                  uint8_t* __restrict__ counts,
                                                                  gf_seqmatch* __restrict__ matches) {
  __shared__ GfMapSmem<LCAP> smem[WAVES];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  GfMapSmem<LCAP>& S = smem[wib];
  const unsigned int nl = *n_list;
  for (unsigned int k = blockIdx.x * WAVES + wib; k < nl; k += gridDim.x * WAVES) {
    const int64_t r = (int64_t)list[k];
    const int64_t off0 = offsets[r];
    const int L = (int)(offsets[r + 1] - off0);
    gf_wave_lds_sync();
    const uint32_t sh = gf_stage_read<LCAP>(S, bases + off0, L, lane);
    gf_wave_lds_sync();
    const int nvotes = gf_first_pass_probe_all<LCAP>(T, S, L, sh, lane);
    gf_finish_read<LCAP>(T, S, L, sh, nvotes, lane, r, counts, matches);
  }
}
