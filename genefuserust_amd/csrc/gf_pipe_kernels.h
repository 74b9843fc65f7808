// Flat ("thread per read") form of Indexer::map_read's first pass for reads of up to 320
// bases — same exact decisions as gf_map_kernels.h, far fewer instructions.
//
// The wave-per-read kernel spends thousands of wave-instructions per read, most of them
// wave-uniform bookkeeping (ballots, scalar branches, LDS staging) that serve one read at a
// time.  Here every lane owns a read, so the same bookkeeping is ordinary per-lane
// arithmetic shared by 64 reads per instruction:
//
//   gf_k_seedverify_stream  thread per read, packing fused in: each wavefront converts the
//                 contiguous bytes of its next 64 reads to 2 bits per base (+ 1 "not A/C/G/T"
//                 bit) straight into its own LDS tile; up to 4 seeds behind the presence
//                 filter, probed one at a time until one names a candidate diagonal K, which
//                 is verified against both strands of the genes laid out in site-code space
//                 (gf_table.h: gdu) with word-parallel bit tricks (16 bases per XOR): window i
//                 counts for K iff its 16 bases equal the bases of site K+i and that site is
//                 the only site of its key.  Then the exact bound ("a diagonal gets at most
//                 one vote per window that can still vote"):
//                   v1 + open < 20 or open < 10  ->  []   (decided, nothing probed)
//                 A read without a candidate goes through the presence filter right here
//                 (two windows per look-up, L2 hits); with P windows left, P < 20 -> [].
//                 Undecided reads are appended, with their packed words, to a list.
//   gf_k_probe_filter  thread per undecided read that has a candidate: the same filter pass
//                 over its unverified windows: v1 + P < 20 or v2 + P < 10 -> []; what is left
//                 is compacted in place.
//   gf_k_probe_buckets  thread per remaining read: probes the P windows' buckets (one 64-byte
//                 bucket each), h = windows that voted; stops as soon as
//                 v1 + h + left < 20 or v2 + h + left < 10 -> [].
//   gf_k_map_reads_list  survivors (junction reads, repeats) and reads beyond 320 bases: the
//                 exact wave-per-read kernel recomputes the read from scratch — votes, top
//                 two, gate, second pass, segments.
// Every read that ends here with [] was *proved* to fail the gate of
// indexer.rs:353-360; everything else is computed by the exact kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"
#include "gf_map_kernels.h"
#include "gf_table.h"

#ifndef GF_PROBE_NT
#define GF_PROBE_NT false
#endif

// 16 flag bits -> the same flags at the even bit positions of a word (2-bit layout)
__device__ __forceinline__ uint32_t gf_spread16(uint32_t x) {
  x &= 0xFFFFu;
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

// every 4th bit (bits 0,4,..,28) of x gathered into the low 8 bits
__device__ __forceinline__ uint32_t gf_gather_nibble_lsb(uint32_t x) {
  x &= 0x11111111u;
  x = (x | (x >> 3)) & 0x03030303u;
  x = (x | (x >> 6)) & 0x000F000Fu;
  x = (x | (x >> 12)) & 0xFFu;
  return x;
}

// wave-aggregated append: returns this lane's slot when `want`, one atomic per wave
__device__ __forceinline__ unsigned int gf_wave_append(bool want, unsigned int* counter) {
  const uint64_t m = __ballot(want);
  unsigned int base = 0;
  if (m) {
    const int leader = __builtin_ctzll(m);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
  }
  return base + (unsigned int)gf_lanes_below(m);
}

// block-local variant: the counter lives in LDS, slots are relative to the block's region
__device__ __forceinline__ unsigned int gf_wave_append_lds(bool want, unsigned int* s_counter) {
  const uint64_t m = __ballot(want);
  unsigned int base = 0;
  if (m) {
    const int leader = __builtin_ctzll(m);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(s_counter, (unsigned int)__popcll(m));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
  }
  return base + (unsigned int)gf_lanes_below(m);
}

// Packing is folded into seed+verify through LDS.  Consecutive reads are consecutive bytes
// (read r = bases[offsets[r] .. offsets[r+1])), so the next 64 reads of a wavefront are one
// contiguous span: it is converted with coalesced 16-byte loads straight into LDS, and each
// thread cuts its read's words out of LDS.  (A separate packing kernel cost the packed
// stream's round trip through HBM, 1.1 GB written + read per 20 M reads, and ~17 dword
// gathers per read from it: L2 hits, but every one of them a request against the ~270 G/s L2
// ceiling that the presence filter also lives on.)  An undecided read's packed words travel
// to the later passes inside its list entry.
// One undecided read: w[0] = read index in the batch, w[1] = v1 | v2 << 8, w[2 .. 2+NT) = one
// bit per stride-2 window still to be asked (bit b of word k = window 32k + b), then the
// read's PW words of 2-bit codes; padded to whole 16-byte vectors.
#define GF_ENTRY_FILTERED 0x80000000u  // in w[1]: the windows listed have already passed the presence filter

template <int PW>
struct GfPipeEntryW {  // 64 B (PW = 10), 96 B (PW = 16), 112 B (PW = 20)
  static constexpr int NT = PW <= 16 ? 4 : (PW + 3) / 4;  // 8 windows per word of the read
  static constexpr int EW = (2 + NT + PW + 3) / 4;        // 16-byte vectors
  uint32_t w[4 * EW];
};

template <int PW>
__device__ __forceinline__ void gf_entry_load(const GfPipeEntryW<PW>* e, uint32_t& r, uint32_t& v1v2,
                                              uint32_t (&m)[GfPipeEntryW<PW>::NT], uint32_t (&pk)[PW + 1]) {
  constexpr int NT = GfPipeEntryW<PW>::NT, EW = GfPipeEntryW<PW>::EW;
  uint32_t w[4 * EW];
  // (non-temporal loads / stores of the entries — they are written once and read once — were measured on
  // IDX-C, where the filter half in use shares the L2 with them: 1.61 against 1.57 ms for the two filter
  // launches; GF_ENTRY_NT builds that form)
  const gf_u32x4* src = (const gf_u32x4*)e;
#pragma unroll
  for (int j = 0; j < EW; ++j) {
#ifdef GF_ENTRY_NT
    const gf_u32x4 q = __builtin_nontemporal_load(src + j);
#else
    const gf_u32x4 q = src[j];
#endif
    w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
  }
  r = w[0];
  v1v2 = w[1];
#pragma unroll
  for (int k = 0; k < NT; ++k) m[k] = w[2 + k];
#pragma unroll
  for (int j = 0; j < PW; ++j) pk[j] = w[2 + NT + j];
  pk[PW] = 0;
}

template <int PW>
__device__ __forceinline__ void gf_entry_store(GfPipeEntryW<PW>* e, uint32_t r, uint32_t v1v2,
                                               const uint32_t (&m)[GfPipeEntryW<PW>::NT], const uint32_t (&pk)[PW + 1]) {
  constexpr int NT = GfPipeEntryW<PW>::NT, EW = GfPipeEntryW<PW>::EW;
  uint32_t w[4 * EW];
  w[0] = r;
  w[1] = v1v2;
#pragma unroll
  for (int k = 0; k < NT; ++k) w[2 + k] = m[k];
#pragma unroll
  for (int j = 0; j < PW; ++j) w[2 + NT + j] = pk[j];
#pragma unroll
  for (int j = 2 + NT + PW; j < 4 * EW; ++j) w[j] = 0;
  gf_u32x4* dst = (gf_u32x4*)e;  // the entry is a multiple of 16 bytes and 16-byte aligned
#pragma unroll
  for (int j = 0; j < EW; ++j) {
    gf_u32x4 q;
    q.x = w[4 * j]; q.y = w[4 * j + 1]; q.z = w[4 * j + 2]; q.w = w[4 * j + 3];
#ifdef GF_ENTRY_NT
    __builtin_nontemporal_store(q, dst + j);
#else
    dst[j] = q;
#endif
  }
}

// ---- K_seedverify, fused with packing (default) ----
// Each wavefront stages its own groups of 64 reads (no block barrier in the loop).  A read's
// words are never held in registers all at once: they stay in the wave's LDS tile and are
// cut out word by word wherever they are needed (clean-window bits, seeds, verification,
// list entry), so the per-lane state is a few rolling words.
__device__ __forceinline__ uint32_t gf_cut_pk(const uint32_t* s_pk, uint32_t w0, uint32_t sh, int j) {
  return __builtin_amdgcn_alignbit(s_pk[w0 + j + 1], s_pk[w0 + j], sh);
}

// unusable-base flags of bases 16j .. 16j+15 of the read, at the even bits (2-bit layout)
__device__ __forceinline__ uint32_t gf_cut_iv(const uint32_t* s_iv, uint32_t pos, int L, int j) {
  const uint32_t bp = pos + 16u * (uint32_t)j;
  uint32_t b = __builtin_amdgcn_alignbit(s_iv[(bp >> 5) + 1], s_iv[bp >> 5], bp & 31u) & 0xFFFFu;
  const int k = L - 16 * j;  // bases at or beyond the end of the read are unusable
  if (k < 16) b |= k <= 0 ? 0xFFFFu : ((0xFFFFu << k) & 0xFFFFu);
  return gf_spread16(b);
}

// bit 2p set iff bases p .. p+15 are all good, for the 16 bases of the word whose bad-base
// flags are z_lo; z_hi = the flags of the next word
__device__ __forceinline__ uint32_t gf_clean16(uint32_t z_lo, uint32_t z_hi) {
  uint32_t lo = ~z_lo & 0x55555555u, hi = ~z_hi & 0x55555555u;
#pragma unroll
  for (uint32_t sh = 2; sh <= 16; sh <<= 1) {
    lo &= __builtin_amdgcn_alignbit(hi, lo, sh);
    hi &= hi >> sh;
  }
  return lo;
}

// 16 ASCII bases -> 32 code bits and 16 "not A/C/G/T" bits.  The expected letter of each
// 2-bit code comes from one v_perm_b32 table lookup; a chunk whose 16 bytes all equal their
// expected letters (the usual case) is done without looking at single bytes.
__device__ __forceinline__ void gf_convert16(const uint4& q, uint32_t& code32, uint32_t& bad16) {
  // twice the code of each base, where it stands in its byte (bits 1, 2): one AND per dword serves both the table
  // look-up of the expected letter and the packing (r04: the kernel is bound by instruction issue, and this runs for
  // every 16 bases of every read — the shifts that made proper codes of them first were four instructions too many)
  const uint32_t z0 = q.x & 0x06060606u, z1 = q.y & 0x06060606u, z2 = q.z & 0x06060606u, z3 = q.w & 0x06060606u;
  // v_perm_b32 with selectors 0, 2, 4, 6: bytes 0 / 2 of the second operand are 'A' / 'C', bytes 0 / 2 of the first 'T' / 'G'
  const uint32_t d0 = q.x ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z0), d1 = q.y ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z1);
  const uint32_t d2 = q.z ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z2), d3 = q.w ^ __builtin_amdgcn_perm(0x00470054u, 0x00430041u, z3);
  // codes of 4 bases -> 8 bits: one dot product of the four bytes with (1, 4, 16, 64) — of the doubled codes, so every
  // sum is twice what it should be: three of them are added up where they belong and halved together
  const uint32_t e0 = __builtin_amdgcn_udot4(z0, 0x40100401u, 0u, false), e1 = __builtin_amdgcn_udot4(z1, 0x40100401u, 0u, false);
  const uint32_t e2 = __builtin_amdgcn_udot4(z2, 0x40100401u, 0u, false), e3 = __builtin_amdgcn_udot4(z3, 0x40100401u, 0u, false);
  code32 = ((e0 + (e1 << 8) + (e2 << 16)) >> 1) | ((e3 >> 1) << 24);
  bad16 = 0;
  if (d0 | d1 | d2 | d3) {
    // bit 7 of each byte = that byte differs; gathered to one bit per base
    const uint32_t n0 = (((d0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d0) & 0x80808080u;
    const uint32_t n1 = (((d1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d1) & 0x80808080u;
    const uint32_t n2 = (((d2 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d2) & 0x80808080u;
    const uint32_t n3 = (((d3 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d3) & 0x80808080u;
    // bits 7,15,23,31 -> bits 0..3
    bad16 = (((n0 >> 7) * 0x00204081u) >> 21 & 0xFu) | ((((n1 >> 7) * 0x00204081u) >> 21 & 0xFu) << 4) |
            ((((n2 >> 7) * 0x00204081u) >> 21 & 0xFu) << 8) | ((((n3 >> 7) * 0x00204081u) >> 21 & 0xFu) << 12);
  }
}

// windows 32k .. 32k+31 of a read with nwin windows, one bit each
__device__ __forceinline__ uint32_t gf_window_mask(int nwin, int k) {
  const int m = nwin - 32 * k;
  return m >= 32 ? 0xFFFFFFFFu : (m <= 0 ? 0u : ((1u << m) - 1u));
}

// which forms of the filter the seeds ask before their bucket probe: 3 = the L2-resident one (2) and the one beyond the
// L2 (1); 2 = only the former.  A look-up of a filter that does not live in the L2 is a miss like the bucket probe it
// is meant to spare — and the probe names the diagonal (r03: IDX-C is bound by missed lines, §5).
#ifndef GF_SEED_FILTER_MASK
#define GF_SEED_FILTER_MASK 2u
#endif
#ifndef GF_MID_SEEDS
#define GF_MID_SEEDS 2  // seeds probed by seed+verify when the filter is not L2-resident (4 = all, as before r03)
// (r03, IDX-C, ms per 20 M reads: PANEL 3.78 with four seeds, 3.50 with two, 3.68 with one; all-background 6.18 / 5.54 /
//  5.27 — a background read pays a missed line per seed, an on-target read whose seeds name nothing pays the filter
//  sweeps and two or three probes in gf_k_probe_buckets)
#endif
#ifndef GF_FILTER_AUX
#define GF_FILTER_AUX 0  // cache policy bits of the inline filter's buffer loads (experiments: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef GF_SVS_WAVES_PER_SIMD
#define GF_SVS_WAVES_PER_SIMD 4  // (four blocks per CU run anyway, see launch_flat: the registers of six are not needed)
#endif
// ---- the background reads' filter rounds, by queue (r04) ----
// r04 measured what binds seed+verify: nine more vector instructions per filter look-up (no memory access) cost 14 % of
// its time — the kernel is bound by instruction issue, and two thirds of its instructions were the two filter rounds
// and their vote bound, executed by every wave for the 40 % of its lanes that hold a background read (round 0) and the
// 24 % that outlive it (round 1).  So the rounds no longer run where a read happens to sit: a read without a candidate
// diagonal is written — its packed words, its standing windows, the round it is due — to a queue in its wave's LDS
// (the layout of a list entry, 64 bytes), and whenever 64 records wait, the wave runs ONE round for 64 of them, every
// lane busy, each on its own record's round.  A record that outlives round 0 goes back into the queue for round 1;
// one that outlives round 1 goes to the list as before; the others are decided ([]).  Same decisions, read by read.
#ifndef GF_FQ_CAP
#define GF_FQ_CAP 96  // records per wave: a tile adds at most 64 to at most 63 waiting (more: passes run first)
#endif

template <int PW>
__device__ __forceinline__ int gf_filter_queue_pass(const GfTable& T, gf_u32x4* s_q, int q_len, int lane,
                                                    uint8_t* __restrict__ counts, GfPipeEntryW<PW>* my_list,
                                                    unsigned int* s_cnt) {
  static_assert(PW == 10, "the queue holds the 64-byte entries of reads of up to 160 bases");
  constexpr int NT = GfPipeEntryW<PW>::NT;
  const int m = q_len < 64 ? q_len : 64;
  const int base = q_len - m;  // the last m records: what a pass leaves behind goes back to the same place
  const bool act = lane < m;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = 0;
  if (act) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gf_u32x4 q = s_q[(base + lane) * 4 + j];
      w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
    }
  }
  gf_wave_lds_sync();  // every lane holds its record: the places may be written again
  const __amdgpu_buffer_rsrc_t filter_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)T.bloom, 0, (int)(T.bloom_words * 4u), 0x00020000);
  const uint32_t filter_bytes = T.bloom_words * 4u;
  const uint32_t rnd = w[1];  // 0: the even pairs of windows are due, 1: the odd pairs
  bool alive = false;
  if (act) {
    uint32_t wlo = w[2 + NT];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const uint32_t whi = j + 1 < PW ? w[2 + NT + j + 1] : 0u;
      const uint32_t byte = (w[2 + (j >> 2)] >> (8 * (j & 3))) & 0xFFu;  // this word's 8 windows
      uint32_t word[2], bits[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // pair u = 2t + rnd of the word: windows 8j+2u, 8j+2u+1 share the 14-mer at bases 16j + 4u+2 ..
        const uint32_t s14 = __builtin_amdgcn_alignbit(whi, wlo, 16u * (uint32_t)t + 8u * rnd + 4u) & 0x0FFFFFFFu;
        const uint32_t h2 = GF_BLOOM_HASH((s14));
        bits[t] = GF_BLOOM_BITS(h2);
        const uint32_t nb = (byte & (3u << (4 * t + 2 * rnd))) ? filter_bytes : 0u;  // nobody to ask for: word 0
        word[t] = __builtin_amdgcn_raw_buffer_load_b32(filter_rsrc, __umulhi(h2, nb) & ~3u, 0, GF_FILTER_AUX);
      }
      uint32_t fail2 = 0;  // bit 2u: the filter rules out pair u's 14-mer
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t d = bits[t] & ~word[t];
        uint32_t f;
        asm("v_min_u32 %0, %1, 1" : "=v"(f) : "v"(d));
        fail2 |= f << (4 * t);
      }
      fail2 <<= 2 * rnd;
      w[2 + (j >> 2)] &= ~((fail2 | (fail2 << 1)) << (8 * (j & 3)));
      wlo = whi;
    }
    uint32_t x[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) x[k] = (w[2 + k] | (w[2 + k] >> 1)) & 0x55555555u;
    alive = gf_vote_bound_pairs<4 * PW - 3>(x) >= GF_MAJOR_KEYS / 2;
  }
  if (act && !alive) counts[w[0]] = 0;  // proved unable to reach the gate (indexer.rs:353-360)
  // round 0 outlived: back into the queue, due for round 1
  const bool again = act && alive && rnd == 0u;
  const uint64_t am = __ballot(again);
  if (again) {
    const int slot = base + gf_lanes_below(am);
    w[1] = 1u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gf_u32x4 q;
      q.x = w[4 * j]; q.y = w[4 * j + 1]; q.z = w[4 * j + 2]; q.w = w[4 * j + 3];
      s_q[slot * 4 + j] = q;
    }
  }
  // both rounds outlived: an undecided read whose listed windows have been through the filter
  const bool emit = act && alive && rnd != 0u;
  const unsigned int slot_b = gf_wave_append_lds(emit, s_cnt);
  if (emit) {
    w[1] = GF_ENTRY_FILTERED;
    gf_u32x4* dst = (gf_u32x4*)(my_list + slot_b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gf_u32x4 q;
      q.x = w[4 * j]; q.y = w[4 * j + 1]; q.z = w[4 * j + 2]; q.w = w[4 * j + 3];
      dst[j] = q;
    }
  }
  gf_wave_lds_sync();
  return base + __popcll(am);
}

// ---- reads with a HIGH seed, by the same queue's other end (r04) ----
// A read whose seeds name no diagonal but one of which is in the table with six sites or more lies, most likely, inside a
// repeat: every window the filter is asked about is present, every bucket probed comes back HIGH.  Such a key keeps a
// representative site (gf_k_index_side) and every site of such a key a flag: the read is verified against the genes on
// the representative's diagonal like an on-target read on its candidate's — windows that equal a site which is the only
// one of its key are votes for that diagonal (v1), windows that equal a flagged site are PROVEN unable to vote (their
// key is that site's key) — and with `open` = the clean windows that are neither, the usual test decides it:
// v1 + open < 20 or open < 10 -> [].  These reads are a fraction of a per cent on SURVEY.md 8(d)'s genes and a fifth
// of all reads on genes with 30 % repeats, so they wait at the BACK of the wave's queue (records: read, diagonal, clean
// windows, packed words) and are verified 64 at a time.
template <int PW>
__device__ __forceinline__ void gf_verify_words2(const GfTable& T, const uint32_t* pk, uint32_t K,
                                                 uint32_t (&vmb)[GfPipeEntryW<PW>::NT], uint32_t (&hmb)[GfPipeEntryW<PW>::NT]) {
  constexpr int NT = GfPipeEntryW<PW>::NT;
#pragma unroll
  for (int k = 0; k < NT; ++k) vmb[k] = hmb[k] = 0;
  const uint2* gp = (const uint2*)T.gdu + (K >> 4);
  if (PW == 10 && T.gdt != nullptr) {
    const uint32_t p = K >> 4, t = __umulhi(p, 0xAAAAAAABu) >> 2;  // t = p / 6
    gp = (const uint2*)T.gdt + 16u * t + (p - 6u * t);
  }
  const uint32_t bo = 2u * (K & 15u);
  uint2 gw[PW + 1];
#pragma unroll
  for (int j = 0; j < PW + 1; ++j) gw[j] = gp[j];
  uint32_t zz_cur;
  {
    const uint32_t x = pk[0] ^ __builtin_amdgcn_alignbit(gw[1].x, gw[0].x, bo);
    zz_cur = (x | (x >> 1)) & 0x55555555u;
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    uint32_t zz_next = 0x55555555u;
    if (j + 1 < PW) {
      const uint32_t x = pk[j + 1] ^ __builtin_amdgcn_alignbit(gw[j + 2 <= PW ? j + 2 : PW].x, gw[j + 1].x, bo);
      zz_next = (x | (x >> 1)) & 0x55555555u;
    }
    const uint32_t u = __builtin_amdgcn_alignbit(gw[j + 1].y, gw[j].y, bo);
    const uint32_t clean = gf_clean16(zz_cur, zz_next);
    vmb[j >> 2] |= gf_gather_nibble_lsb(clean & u & 0x11111111u) << (8 * (j & 3));
    hmb[j >> 2] |= gf_gather_nibble_lsb(clean & (u >> 1) & 0x11111111u) << (8 * (j & 3));
    zz_cur = zz_next;
  }
}

template <int PW>
__device__ __forceinline__ int gf_high_queue_pass(const GfTable& T, gf_u32x4* s_q, int q_back, int lane,
                                                  uint8_t* __restrict__ counts, GfPipeEntryW<PW>* my_list,
                                                  unsigned int* s_cnt) {
  static_assert(PW == 10, "the queue holds the 64-byte entries of reads of up to 160 bases");
  constexpr int NT = GfPipeEntryW<PW>::NT;
  const int m = q_back < 64 ? q_back : 64;
  const int base = GF_FQ_CAP - q_back;  // the records added last (the back end grows downwards)
  const bool act = lane < m;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = 0;
  if (act) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gf_u32x4 q = s_q[(base + lane) * 4 + j];
      w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
    }
  }
  bool emit = false;
  if (act) {
    uint32_t vk[NT], hk[NT];
    gf_verify_words2<PW>(T, &w[2 + NT], w[1], vk, hk);  // (w[2 + NT + PW] does not exist: the function reads PW words)
    int v1 = 0, h = 0, nvalid = 0;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const uint32_t cw = w[2 + k];  // clean windows (the words past the read's end compared garbage: masked here)
      vk[k] &= cw;
      hk[k] &= cw & ~vk[k];
      v1 += __popc(vk[k]);
      h += __popc(hk[k]);
      nvalid += __popc(cw);
      w[2 + k] = cw & ~vk[k] & ~hk[k];  // what may still vote for another diagonal
    }
    const int open = nvalid - v1 - h;
    if (v1 + open < GF_MAJOR_KEYS / 2 || open < GF_MINOR_KEYS / 2) counts[w[0]] = 0;
    else emit = true;
    // v1 > 0: an entry with a candidate diagonal (counted windows are votes); v1 == 0: nobody need ask the filter about
    // windows of a read that lies in the table all over — marked filtered, the bucket pass takes it from here
    w[1] = v1 > 0 ? (uint32_t)v1 : GF_ENTRY_FILTERED;
  }
  const unsigned int slot_b = gf_wave_append_lds(emit, s_cnt);
  if (emit) {
    gf_u32x4* dst = (gf_u32x4*)(my_list + slot_b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gf_u32x4 q;
      q.x = w[4 * j]; q.y = w[4 * j + 1]; q.z = w[4 * j + 2]; q.w = w[4 * j + 3];
      dst[j] = q;
    }
  }
  gf_wave_lds_sync();
  return q_back - m;
}

// PACKED: the reads arrive as the 2-bit + bad-bit form of the whole `bases` stream (gf_pack_bases_device:
// word c of g_pk / g_iv = bases 16c .. 16c+15, `offsets` still count bases) — a tile is then copied,
// 6 bytes per 16 bases instead of 16, and nothing is converted.
template <int PW, bool PACKED = false>
__global__ __launch_bounds__(256, PW <= 10 ? GF_SVS_WAVES_PER_SIMD : 4) void gf_k_seedverify_stream(
    GfTable T, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ g_pk,
    const uint16_t* __restrict__ g_iv, const int64_t* __restrict__ offsets, int64_t n, int lmax,
    int batch_max, uint8_t* __restrict__ counts, GfPipeEntryW<PW>* __restrict__ list_b,
    unsigned int* __restrict__ blk_cnt, int64_t per_block, uint32_t* __restrict__ list_long,
    unsigned int* __restrict__ ctr) {
  // lmax = longest read this kernel maps; reads of lmax+1 .. batch_max bases are handed to the
  // wave-per-read kernels of the longer classes through list_long (<= 1024 bases from the
  // front, counter ctr[2]; longer from the back, counter ctr[3]); beyond batch_max: marked.
  constexpr int TILE_BYTES = 64 * 16 * PW;         // ASCII bytes staged per tile (64 reads of 16*PW bases)
  constexpr int TILE_CHUNKS = TILE_BYTES / 16 + 1;  // +1: the span starts at a 16-byte boundary at or below its first read
  constexpr int PK_WORDS = (TILE_CHUNKS + PW + 2 + 3) & ~3;  // (whole 16-byte vectors: the tile doubles as the entries' staging area)
  constexpr int IV_WORDS = (TILE_CHUNKS + PW + 2) / 2 + 2;
  constexpr int NLOAD = (TILE_CHUNKS + 63) / 64;    // 16-byte chunks per lane per tile
  constexpr int IW = (PW + 1) / 2;                  // 32-base words of flag bits per read
  constexpr int NT = GfPipeEntryW<PW>::NT;          // words of one bit per stride-2 window
  __shared__ __attribute__((aligned(16))) uint32_t s_pk_all[4][PK_WORDS];
  __shared__ uint32_t s_iv_all[4][IV_WORDS];
#ifndef GF_SV_INLINE_ROUNDS
  constexpr bool QUEUED = PW == 10;  // the filter rounds of reads without a candidate diagonal go by queue (above)
#else
  constexpr bool QUEUED = false;     // (A/B build: the rounds inline, as in r03)
#endif
  __shared__ gf_u32x4 s_q_all[QUEUED ? 4 : 1][QUEUED ? GF_FQ_CAP * 4 : 1];
  __shared__ unsigned int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* s_pk = s_pk_all[wave];
  uint32_t* s_iv = s_iv_all[wave];
  gf_u32x4* s_q = s_q_all[QUEUED ? wave : 0];
  int q_len = 0;   // records waiting at the front of this wave's queue: filter rounds (wave-uniform)
  int q_back = 0;  // ... and at its back: reads with a HIGH seed
  const int64_t r_lo = (int64_t)blockIdx.x * per_block;
  int64_t n_lim = n;
  if (T.n_dev) {  // (the reads beyond the device-side count are empty slots nobody asks about)
    const int64_t nd = *T.n_dev;
    n_lim = nd < n ? (nd < 0 ? 0 : nd) : n;
    // (Sharing the reads that exist evenly among all the blocks made this kernel faster — 0.39 -> 0.33 ms for 1.6 M
    //  merged reads in 10 M slots — and the two list kernels behind it slower by more, 0.19 -> 0.42 ms: they pay per
    //  block that holds an entry, and 6250 blocks then held a few each instead of 1325 many.)
  }
  const int64_t r_hi = r_lo + per_block < n_lim ? r_lo + per_block : n_lim;
  const int64_t fixed_len = T.fixed_len;
  GfPipeEntryW<PW>* my_list = list_b + r_lo;
  for (int64_t g0 = r_lo + 64 * (int64_t)wave; g0 < r_hi; g0 += 256) {
    const int64_t g1 = g0 + 64 < r_hi ? g0 + 64 : r_hi;
    int64_t r0 = g0;
    while (r0 < g1) {
      // the reads r0 .. r0+nfit-1 (a prefix of the group) fit in the tile.  The tile starts at
      // the 16-byte boundary at or below the first read (pointer arithmetic on `bases` keeps
      // the loads in the global address space with a scalar base).
#define GF_OFF(x) (fixed_len ? (int64_t)(x) * fixed_len : offsets[x])  // (fixed_len is wave-uniform: a scalar branch)
      const int64_t base_off = GF_OFF(r0);
      const uint8_t* p0 = bases + base_off;
      const uint32_t mis = PACKED ? (uint32_t)(base_off & 15) : (uint32_t)((uintptr_t)p0 & 15u);
      const uint4* src = (const uint4*)(p0 - mis);
      const int64_t c0 = base_off >> 4;  // (PACKED) first chunk of the tile in the packed stream
      const int64_t r = r0 + lane;
      int64_t off0 = 0, off1 = 0;
      if (r < g1) {
        off0 = GF_OFF(r);
        off1 = GF_OFF(r + 1);
      }
      const bool fits = r < g1 && (uint64_t)(off1 - base_off) + mis <= (uint64_t)TILE_BYTES;
      int nfit = __popcll(__ballot(fits));
      const bool oversize = nfit == 0;  // a single read larger than the tile: far beyond lmax, nothing to stage
      if (oversize) nfit = 1;
      gf_wave_lds_sync();  // the previous tile's LDS reads are done
      const uint32_t chunks = oversize ? 0u : (uint32_t)((GF_OFF(r0 + nfit) - base_off) + mis + 15) >> 4;
      if (PACKED && chunks > 0) {
        uint32_t qp[NLOAD], qi[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {  // all of the tile's loads in flight together
          const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
          const int64_t cc = c0 + (int64_t)(c < chunks ? c : chunks - 1);
          qp[k] = __builtin_nontemporal_load(g_pk + cc);
          qi[k] = __builtin_nontemporal_load(g_iv + cc);
        }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
          const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
          if (c < chunks) {
            s_pk[c] = qp[k];
            ((uint16_t*)s_iv)[c] = (uint16_t)qi[k];
          }
        }
      } else if (chunks > 0) {  // (empty reads only: nothing to stage)
        uint4 q[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {  // all of the tile's loads in flight together
          const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
#if defined(GF_STAGE_AUX)  // experiments: the staging loads as buffer loads with explicit cache-policy bits (1 sc0, 2 nt, 16 sc1)
          const __amdgpu_buffer_rsrc_t stage_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(chunks * 16u), 0x00020000);
          const auto t = __builtin_amdgcn_raw_buffer_load_b128(stage_rsrc, 16u * (c < chunks ? c : chunks - 1), 0, GF_STAGE_AUX);
          q[k] = make_uint4(t[0], t[1], t[2], t[3]);
#elif !defined(GF_STAGE_TEMPORAL)  // streamed once: keep the bases out of the way of the table and the filter
          const gf_u32x4 t = __builtin_nontemporal_load((const gf_u32x4*)(src + (c < chunks ? c : chunks - 1)));
          q[k] = make_uint4(t.x, t.y, t.z, t.w);
#else
          q[k] = src[c < chunks ? c : chunks - 1];
#endif
        }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
          const uint32_t c = (uint32_t)lane + 64u * (uint32_t)k;
          uint32_t code32, bad16;
          gf_convert16(q[k], code32, bad16);
          if (c < chunks) {
            s_pk[c] = code32;
            ((uint16_t*)s_iv)[c] = (uint16_t)bad16;
          }
        }
      }
      gf_wave_lds_sync();
      __builtin_amdgcn_sched_barrier(0);
      const bool in_range = lane < nfit;
      bool undecided = false, long1k = false, long4k = false;
      bool queued = false;  // a read without a candidate diagonal, on its way to the wave's filter queue
      bool hqueued = false; // ... or, with a HIGH seed, to the queue's other end
      uint32_t K_q = 0;     // (the diagonal it is verified on there)
      uint32_t pp_q[NT];    // ... and its standing / clean windows
#pragma unroll
      for (int k = 0; k < NT; ++k) pp_q[k] = 0;
      uint32_t e_v1v2 = 0, e_todo[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) e_todo[k] = 0;
      uint32_t w0 = 0, sh = 0;
#if defined(GF_ABLATE_SV) && GF_ABLATE_SV == 6  // timing only: staging and conversion, one LDS word per read
      if (in_range) counts[r] = s_pk[((uint32_t)(off0 - base_off) + mis) >> 4] == 0x1234567u ? 1 : 0;
      if (false) {
#else
      if (in_range) {
#endif
        const int64_t len64 = off1 - off0;
        if (T.skip != nullptr && T.skip[r] > 0) {
          counts[r] = 0;  // not a candidate of this pass (gf_table.h: skip)
        } else if (len64 > batch_max) {
          counts[r] = GF_COUNT_TOO_LONG;
        } else if (len64 > lmax) {
          long1k = len64 <= 1024;
          long4k = !long1k;
        } else if (len64 < GF_KMER + 2 * (GF_MAJOR_KEYS / 2 - 1)) {
          counts[r] = 0;  // fewer than 20 stride-2 windows: count1 < 20 whatever they hit
        } else {
          const int L = (int)len64;
          const uint32_t pos = (uint32_t)(off0 - base_off) + mis;
          w0 = pos >> 4;
          sh = 2u * (pos & 15u);
          // does the read hold any base outside A/C/G/T?  (flag bits pos .. pos+L-1)
          uint32_t anybad = 0;
          {
            const uint32_t v0 = pos >> 5, vs = pos & 31u;
            uint32_t lo = s_iv[v0];
#pragma unroll
            for (int j = 0; j < IW; ++j) {
              const uint32_t hi = s_iv[v0 + j + 1];
              uint32_t b = __builtin_amdgcn_alignbit(hi, lo, vs);
              const int k = L - 32 * j;
              if (k < 32) b &= k <= 0 ? 0u : ((1u << k) - 1u);
              anybad |= b;
              lo = hi;
            }
          }
          // pass A: the clean stride-2 windows (all 16 bases usable), one bit per window, and
          // the seeds at bases 0, 32, 64, 96
          uint32_t cwb[NT];
          int nvalid;
          uint32_t key[4];
          uint32_t okm = 0;
#pragma unroll
          for (int s = 0; s < 4; ++s) key[s] = 2 * s < PW ? gf_cut_pk(s_pk, w0, sh, 2 * s) : 0u;
          if (!anybad) {
            nvalid = (L - GF_KMER) / 2 + 1;
#pragma unroll
            for (int k = 0; k < NT; ++k) cwb[k] = gf_window_mask(nvalid, k);
#pragma unroll
            for (int s = 0; s < 4; ++s) okm |= (2 * s < PW && 16 * s < nvalid ? 1u : 0u) << s;
          } else {
#pragma unroll
            for (int k = 0; k < NT; ++k) cwb[k] = 0;
            nvalid = 0;
            uint32_t z_cur = gf_cut_iv(s_iv, pos, L, 0);
#pragma unroll
            for (int j = 0; j < PW; ++j) {
              const uint32_t z_next = j + 1 < PW ? gf_cut_iv(s_iv, pos, L, j + 1) : 0x55555555u;
              const uint32_t cw = gf_clean16(z_cur, z_next) & 0x11111111u;
              nvalid += __popc(cw);
              cwb[j >> 2] |= gf_gather_nibble_lsb(cw) << (8 * (j & 3));
              if ((j & 1) == 0 && j < 8) okm |= (cw & 1u) << (j >> 1);
              z_cur = z_next;
            }
          }
#if defined(GF_ABLATE_SV) && GF_ABLATE_SV == 1
          if ((nvalid ^ key[0] ^ key[1] ^ key[2] ^ key[3] ^ cwb[0] ^ cwb[1] ^ cwb[2] ^ cwb[NT - 1] ^ okm) == 0x1234567u) counts[r] = 1;
          okm = 0;
#endif
#if defined(GF_ABLATE_SV) && GF_ABLATE_SV == 5  // timing only: staging, conversion, the windows and seeds of every read
          if ((nvalid ^ key[0] ^ key[1] ^ key[2] ^ key[3] ^ cwb[0] ^ cwb[1] ^ cwb[2] ^ cwb[NT - 1] ^ okm) == 0x1234567u) counts[r] = 1;
          okm = 0;
          nvalid = 0;
#pragma unroll
          for (int k = 0; k < NT; ++k) cwb[k] = 0;
#endif
          // The seeds go through the presence filter before their buckets (L2 hits).  The 14-mer
          // asked about (bases 32s+2 .. 32s+15) is the last 14 bases of window 16s and the first 14
          // of window 16s+1: a clear bit pair proves that neither can vote, which spares the filter
          // pass those windows.
          uint32_t kill[2] = {0, 0};  // windows 0..63 proven unable to vote (seeds sit at windows 0, 16, 32, 48)
          // ... of which these are IN the table with six sites or more (HIGH): they cannot vote, but for the vote bound
          // (gf_table.h) they are not "absent" — a HIGH window may well sit between two voters of one diagonal, and
          // striking it would split their run (r03: found by enumeration, tests/test_vote_bound.py)
          uint32_t khigh[2] = {0, 0};
          uint32_t K = GF_NONE_LIN;   // candidate diagonal: site code of read base 0
          uint32_t Krep = GF_NONE_LIN;  // ... of a copy of the repeat the read may lie in (a HIGH seed's representative site)
          // Seed 0 first — filter, then its bucket: most reads that have a candidate diagonal get it
          // here, for one filter line into the L1 instead of four (the kernel is bound by those line
          // fills); the other three seeds are asked about only by the reads still without one.
          if (T.bloom_in_l2 & GF_SEED_FILTER_MASK) {
            const uint32_t h2 = GF_BLOOM_HASH((key[0] >> 4));
            const uint32_t fb = GF_BLOOM_BITS(h2);
            if ((T.bloom[GF_BLOOM_WORD(h2, T.bloom_words)] & fb) != fb) {
              okm &= ~1u;
              kill[0] |= 3u;
            }
          }
          if (okm & 1u) {
            okm &= ~1u;
            const uint32_t val = gf_lookup(T, key[0]);
            const uint32_t ty = val >> GF_TYPE_SHIFT;
            if (ty == GF_TYPE_UNIQUE) K = val & GF_LIN_MASK;
            else if (ty != GF_TYPE_DUPES) {
              kill[0] |= 1u;
              if (ty == GF_TYPE_HIGH) {
                khigh[0] |= 1u;
                if ((val & GF_LIN_MASK) != GF_LIN_MASK) Krep = val & GF_LIN_MASK;  // the key's representative site
              }
            }
          }
          if (K == GF_NONE_LIN && (T.bloom_in_l2 & GF_SEED_FILTER_MASK)) {
            uint32_t fw[4], fb[4];
#pragma unroll
            for (int s = 1; s < 4; ++s) {
              const uint32_t h2 = GF_BLOOM_HASH((key[s] >> 4));
              fb[s] = GF_BLOOM_BITS(h2);
              fw[s] = T.bloom[GF_BLOOM_WORD(h2, T.bloom_words)];
            }
#pragma unroll
            for (int s = 1; s < 4; ++s)
              if (2 * s < PW && (fw[s] & fb[s]) != fb[s]) {
                okm &= ~(1u << s);
                kill[s >> 1] |= 3u << (16 * (s & 1));
              }
          }
#if defined(GF_ABLATE_SV) && GF_ABLATE_SV == 3
          if (okm == 0x1234567u) counts[r] = 1;
          okm = 0;
#endif
          // With a filter that lives beyond the L2 the seeds go to their buckets unasked, and then only GF_MID_SEEDS of
          // them (r03): a background read's four misses were a quarter of IDX-C's seed+verify; a read whose first
          // seeds name nothing gets its diagonal in gf_k_probe_buckets, if it outlives the filter sweeps.
          if (T.bloom_in_l2 != 2) okm &= (1u << GF_MID_SEEDS) - 1u;
          // one bucket probe at a time, in seed order, until one names a diagonal: an
          // on-target read costs one L2-missing request here
          while (okm != 0 && K == GF_NONE_LIN) {  // (a wave runs as many rounds as its unluckiest lane needs)
            const int s = __builtin_ctz(okm);
            okm &= okm - 1;
            const uint32_t ks = s == 0 ? key[0] : (s == 1 ? key[1] : (s == 2 ? key[2] : key[3]));
            const uint32_t val = gf_lookup(T, ks);
            const uint32_t ty = val >> GF_TYPE_SHIFT;
            if (ty == GF_TYPE_UNIQUE) K = (val & GF_LIN_MASK) - 32u * (uint32_t)s;
            else if (ty != GF_TYPE_DUPES) {  // absent or >= 6 sites: no vote
              kill[s >> 1] |= 1u << (16 * (s & 1));
              if (ty == GF_TYPE_HIGH) {
                khigh[s >> 1] |= 1u << (16 * (s & 1));
                if (Krep == GF_NONE_LIN && (val & GF_LIN_MASK) != GF_LIN_MASK) Krep = (val & GF_LIN_MASK) - 32u * (uint32_t)s;
              }
            }
          }
          // windows that cannot vote are not "clean" for what follows
          nvalid -= __popc(cwb[0] & kill[0]) + __popc(cwb[1] & kill[1]);
          cwb[0] &= ~kill[0];
          cwb[1] &= ~kill[1];
#if defined(GF_ABLATE_SV) && GF_ABLATE_SV == 4
          if (K == 0x1234567u) counts[r] = 1;
          K = GF_NONE_LIN;
#endif
          // A read without a candidate diagonal (background, mostly) goes through the presence
          // filter right here: seed+verify is bound by L2-missing requests and leaves the L2's hit
          // bandwidth idle, which is exactly what the filter pass is short of.  Windows 8j .. 8j+7
          // per step, two windows per look-up; stops when even the windows not asked yet cannot
          // reach the gate.  Survivors carry the windows still standing and a flag that tells
          // gf_k_probe_filter to pass them on as they are.
          bool filt_done = false, filt_dead = false;
#ifdef GF_ABLATE_EXTRA_VALU
          uint32_t ablate_sink = 0;
#endif
          const __amdgpu_buffer_rsrc_t filter_rsrc =
              __builtin_amdgcn_make_buffer_rsrc((void*)T.bloom, 0, (int)(T.bloom_words * 4u), 0x00020000);
          const uint32_t filter_bytes = T.bloom_words * 4u;
          uint32_t pp[NT];
#pragma unroll
          for (int k = 0; k < NT; ++k) pp[k] = 0;
#ifndef GF_SV_NO_INLINE_FILTER
#ifndef GF_FILTER_ALLPAIRS
          // r03, reads of up to 160 bases: HALF the look-ups, and the bound of gf_table.h (gf_vote_bound_pairs)
          // in place of "fewer than 20 windows left".  Round 0 asks the even pairs of windows (0,1), (4,5),
          // (8,9) .. — the seeds' pairs, already answered, are among them — which leaves the odd pairs standing,
          // two windows with two ruled-out ones either side: a diagonal then gets at most 2 votes per 12 windows,
          // 12 for a 150-base read where 20 are needed, and a false positive of the filter adds 3.  A read the
          // bound does not stop (three false positives or more, or windows that really are in the table) asks
          // the odd pairs in round 1 and meets the bound again with everything known.  13 + 4 look-ups instead of
          // 28 + 4 for a background read, and these look-ups — their instructions, r04 found — are the kernel's bound.
          if (QUEUED && K == GF_NONE_LIN && T.bloom_in_l2 == 2) {
            // by queue (gf_filter_queue_pass): the standing windows go along; a read with fewer than 20 clean
            // windows left cannot reach the gate whatever the filter says
            filt_done = true;
            if (nvalid < GF_MAJOR_KEYS / 2) {
              filt_dead = true;
            } else if ((khigh[0] | khigh[1]) && Krep != GF_NONE_LIN) {
              // a seed is in the table with six sites or more: most likely a read inside a repeat — verified against
              // the copy of it that the key's representative site names, from the back of the queue (gf_high_queue_pass)
              hqueued = true;
#pragma unroll
              for (int k = 0; k < NT; ++k) pp_q[k] = cwb[k];
              K_q = Krep;
            } else if (khigh[0] | khigh[1]) {
              // ... and no representative to go by (the two-pass build keeps none): the filter will call its every
              // window present — it skips the rounds and goes on as it is; the bucket pass strikes
              // its windows from one probe (gf_k_probe_buckets, r04).  Listing windows the filter was not asked about
              // as "filtered" is sound: the mark only says nobody need ask again.
#pragma unroll
              for (int k = 0; k < NT; ++k) pp[k] = cwb[k];
              pp[0] |= khigh[0];
              pp[1] |= khigh[1];
            } else {
              queued = true;
#pragma unroll
              for (int k = 0; k < NT; ++k) pp_q[k] = cwb[k];
              pp_q[0] |= khigh[0];  // (in the table: they stand, as far as the bound's runs are concerned)
              pp_q[1] |= khigh[1];
            }
          } else if (PW == 10 && K == GF_NONE_LIN && T.bloom_in_l2 == 2) {
            filt_done = true;
#pragma unroll
            for (int k = 0; k < NT; ++k) pp[k] = cwb[k];
            pp[0] |= khigh[0];  // (in the table: they stand, as far as the bound's runs are concerned)
            pp[1] |= khigh[1];
            bool alive = true;
#pragma unroll 1
            for (uint32_t rnd = 0; rnd < 2; ++rnd) {
              if (alive) {
                uint32_t wlo = gf_cut_pk(s_pk, w0, sh, 0);
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                  const uint32_t whi = j + 1 < PW ? gf_cut_pk(s_pk, w0, sh, j + 1) : 0u;
                  const uint32_t byte = (pp[j >> 2] >> (8 * (j & 3))) & 0xFFu;  // this word's 8 windows
                  uint32_t word[2], bits[2];
#pragma unroll
                  for (int t = 0; t < 2; ++t) {
                    // pair u = 2t + rnd of the word: windows 8j+2u, 8j+2u+1 share the 14-mer at bases 16j + 4u+2 ..
                    const uint32_t s14 = __builtin_amdgcn_alignbit(whi, wlo, 16u * (uint32_t)t + 8u * rnd + 4u) & 0x0FFFFFFFu;
                    const uint32_t h2 = GF_BLOOM_HASH((s14));
#ifdef GF_ABLATE_EXTRA_VALU  // timing only: one more canonical form per look-up (8 vector instructions, no memory access)
                    ablate_sink ^= gf_canon14(s14 ^ 0x05A5A5A5u);
#endif
                    bits[t] = GF_BLOOM_BITS(h2);
                    const uint32_t nb = (byte & (3u << (4 * t + 2 * rnd))) ? filter_bytes : 0u;  // nobody to ask for: word 0
                    word[t] = __builtin_amdgcn_raw_buffer_load_b32(filter_rsrc, __umulhi(h2, nb) & ~3u, 0, GF_FILTER_AUX);
                  }
                  uint32_t fail2 = 0;  // bit 2u: the filter rules out pair u's 14-mer
#pragma unroll
                  for (int t = 0; t < 2; ++t) {
                    const uint32_t d = bits[t] & ~word[t];
                    uint32_t f;
                    asm("v_min_u32 %0, %1, 1" : "=v"(f) : "v"(d));
                    fail2 |= f << (4 * t);
                  }
                  fail2 <<= 2 * rnd;
                  pp[j >> 2] &= ~((fail2 | (fail2 << 1)) << (8 * (j & 3)));
                  wlo = whi;
                }
                uint32_t x[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) x[k] = (pp[k] | (pp[k] >> 1)) & 0x55555555u;
                alive = gf_vote_bound_pairs<4 * PW - 3>(x) >= GF_MAJOR_KEYS / 2;
              }
            }
            filt_dead = !alive;
          } else
#endif
          if (K == GF_NONE_LIN && T.bloom_in_l2 == 2) {
            filt_done = true;
            int npos = 0, rem = nvalid;
            uint32_t wlo = gf_cut_pk(s_pk, w0, sh, 0);
#pragma unroll
            for (int j = 0; j < PW; ++j) {
              const uint32_t whi = j + 1 < PW ? gf_cut_pk(s_pk, w0, sh, j + 1) : 0u;
#ifndef GF_FILTER_OLD
              // (r02 e: the kernel's vector ALUs are busy 78 % of the time and this loop is half of their
              //  instructions.  The four answers of a word are gathered as fail bits on the even positions of one
              //  mask and applied to the word's window byte in one go — an and-not, a compare and a shift-or per
              //  look-up where the first form spent a dozen on compares, selects and counters; a look-up nobody
              //  needs reads word 0, a line all lanes share, instead of branching round its load.)
              if (!filt_dead) {
                const uint32_t byte = (cwb[j >> 2] >> (8 * (j & 3))) & 0xFFu;  // this word's 8 windows
                uint32_t word[4], bits[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  // windows 8j+2u and 8j+2u+1 share the 14-mer at bases 16j + 4u+2 .. 16j + 4u+15
                  const uint32_t s14 = __builtin_amdgcn_alignbit(whi, wlo, 8u * (uint32_t)u + 4u) & 0x0FFFFFFFu;
                  const uint32_t h2 = GF_BLOOM_HASH((s14));
                  bits[u] = GF_BLOOM_BITS(h2);
                  // (a buffer load: its 32-bit byte offset is the whole address computation; nwords = 0 for a
                  //  look-up nobody needs sends it to word 0; floor(h * 4n / 2^32) & ~3 = 4 * floor(h * n / 2^32))
                  const uint32_t nb = (byte & (3u << (2 * u))) ? filter_bytes : 0u;
                  word[u] = __builtin_amdgcn_raw_buffer_load_b32(filter_rsrc, __umulhi(h2, nb) & ~3u, 0, GF_FILTER_AUX);
                }
                uint32_t fail2 = 0;  // bit 2u: the filter rules out look-up u's 14-mer
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const uint32_t t = bits[u] & ~word[u];
                  uint32_t f;  // min(t, 1): the compiler turns the C form into a compare and a select
                  asm("v_min_u32 %0, %1, 1" : "=v"(f) : "v"(t));
                  fail2 |= f << (2 * u);
                }
                const uint32_t pbyte = byte & ~(fail2 | (fail2 << 1));
                pp[j >> 2] |= pbyte << (8 * (j & 3));
                npos += __popc(pbyte);
                rem -= __popc(byte);
                filt_dead = npos + rem < GF_MAJOR_KEYS / 2;
              }
#else
              if (!filt_dead) {
                const uint32_t byte = (cwb[j >> 2] >> (8 * (j & 3))) & 0xFFu;  // this word's 8 windows
                uint32_t word[4], bits[4], both[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  both[u] = (byte >> (2 * u)) & 3u;
                  // windows 8j+2u and 8j+2u+1 share the 14-mer at bases 16j + 4u+2 .. 16j + 4u+15
                  const uint32_t s14 = __builtin_amdgcn_alignbit(whi, wlo, 8u * (uint32_t)u + 4u) & 0x0FFFFFFFu;
                  const uint32_t h2 = GF_BLOOM_HASH((s14));
                  bits[u] = GF_BLOOM_BITS(h2);
                  word[u] = 0;
                  if (both[u]) word[u] = T.bloom[GF_BLOOM_WORD(h2, T.bloom_words)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const int cnt = (int)(both[u] & 1u) + (int)(both[u] >> 1);
                  rem -= cnt;
                  if (both[u] && (word[u] & bits[u]) == bits[u]) {
                    pp[j >> 2] |= both[u] << (8 * (j & 3) + 2 * u);
                    npos += cnt;
                  }
                }
                filt_dead = npos + rem < GF_MAJOR_KEYS / 2;
              }
#endif
              wlo = whi;
            }
          }
#endif
          // pass B: verify the candidate diagonal against the genes in site-code space, a
          // word of the read at a time: window w counts iff its 16 bases equal the bases of
          // site K + 2w and that site is the only site of its key
          uint32_t vmb[NT];  // verified windows, one bit per window like cwb
#pragma unroll
          for (int k = 0; k < NT; ++k) vmb[k] = 0;
          if (K != GF_NONE_LIN) {
            const uint2* gp = (const uint2*)T.gdu + (K >> 4);  // (gd word, ub2 word) pairs
            if (PW == 10 && T.gdt != nullptr) {  // the tiled copy: these 11 pairs in one cache line (gf_table.h: gdt)
              const uint32_t p = K >> 4, t = __umulhi(p, 0xAAAAAAABu) >> 2;  // t = p / 6
              gp = (const uint2*)T.gdt + 16u * t + (p - 6u * t);
            }
            const uint32_t bo = 2u * (K & 15u);
            uint2 gw[PW + 1];
#pragma unroll
            for (int j = 0; j < PW + 1; ++j) gw[j] = gp[j];
            uint32_t zz_cur;
            {
              const uint32_t x = gf_cut_pk(s_pk, w0, sh, 0) ^ __builtin_amdgcn_alignbit(gw[1].x, gw[0].x, bo);
              zz_cur = (x | (x >> 1)) & 0x55555555u;  // mismatching base
              if (anybad) zz_cur |= gf_cut_iv(s_iv, pos, L, 0);
            }
#pragma unroll
            for (int j = 0; j < PW; ++j) {
              uint32_t zz_next = 0x55555555u;
              if (j + 1 < PW) {
                const uint32_t x = gf_cut_pk(s_pk, w0, sh, j + 1) ^
                                   __builtin_amdgcn_alignbit(gw[j + 2 <= PW ? j + 2 : PW].x, gw[j + 1].x, bo);
                zz_next = (x | (x >> 1)) & 0x55555555u;
                if (anybad) zz_next |= gf_cut_iv(s_iv, pos, L, j + 1);
              }
              const uint32_t u = __builtin_amdgcn_alignbit(gw[j + 1].y, gw[j].y, bo);
              const uint32_t ver = gf_clean16(zz_cur, zz_next) & u & 0x11111111u;
              vmb[j >> 2] |= gf_gather_nibble_lsb(ver) << (8 * (j & 3));
              zz_cur = zz_next;
              if (j & 1) __builtin_amdgcn_sched_barrier(0);  // keep it a stream: two words in flight
            }
          }
          int v1 = 0;
#ifdef GF_ABLATE_EXTRA_VALU
          if (ablate_sink == 0x12345u) v1 = 1;
#endif
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            vmb[k] &= cwb[k];  // windows that run past the end of the read compared garbage
            v1 += __popc(vmb[k]);
          }
          // every other diagonal gets at most one vote per window that can still vote
          const int open = nvalid - v1;
          if (queued || hqueued) {
            // (decided by the queue's passes)
          } else if (v1 + open < GF_MAJOR_KEYS / 2 || open < GF_MINOR_KEYS / 2 || filt_dead) {
            counts[r] = 0;
          } else if (filt_done) {
            undecided = true;
            e_v1v2 = GF_ENTRY_FILTERED;  // v1 = v2 = 0; the windows below have been through the filter
#pragma unroll
            for (int k = 0; k < NT; ++k) e_todo[k] = pp[k];
          } else if (QUEUED && K != GF_NONE_LIN && (khigh[0] | khigh[1])) {
            // undecided on its diagonal, and a seed of it is HIGH: a read across the edge of a repeat.  The windows
            // inside the repeat equal HIGH-flagged sites on this same diagonal: the queue's back end strikes them
            // (gf_high_queue_pass verifies on K once more, with both kinds of flags) and most such reads end there
            hqueued = true;
            K_q = K;
#pragma unroll
            for (int k = 0; k < NT; ++k) pp_q[k] = cwb[k];
          } else {
            undecided = true;
            e_v1v2 = (uint32_t)v1;  // v2 = 0: one candidate diagonal per read
#pragma unroll
            for (int k = 0; k < NT; ++k) e_todo[k] = cwb[k] & ~vmb[k];
            if (v1 == 0) {  // goes by the bound in gf_k_probe_filter: HIGH seeds stand there too
              e_todo[0] |= khigh[0];
              e_todo[1] |= khigh[1];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (QUEUED) {
        const uint64_t qm = __ballot(queued), hm = __ballot(hqueued);
        if (qm | hm) {
          const int n_new = __popcll(qm), n_newh = __popcll(hm);
          while (q_len + q_back + n_new + n_newh > GF_FQ_CAP) {  // (rare: room is made at the fuller end)
            if (q_len >= q_back) q_len = gf_filter_queue_pass<PW>(T, s_q, q_len, lane, counts, my_list, &s_cnt);
            else q_back = gf_high_queue_pass<PW>(T, s_q, q_back, lane, counts, my_list, &s_cnt);
          }
          if (queued || hqueued) {
            // the record = the read's list entry; its second word the round it is due (front) or its diagonal (back)
            const int slot = queued ? q_len + gf_lanes_below(qm) : GF_FQ_CAP - 1 - q_back - gf_lanes_below(hm);
            uint32_t ew[16];
            ew[0] = (uint32_t)r;
            ew[1] = queued ? 0u : K_q;
#pragma unroll
            for (int k = 0; k < NT; ++k) ew[2 + k] = pp_q[k];
#pragma unroll
            for (int j = 0; j < PW; ++j) ew[2 + NT + j] = gf_cut_pk(s_pk, w0, sh, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              gf_u32x4 q;
              q.x = ew[4 * j]; q.y = ew[4 * j + 1]; q.z = ew[4 * j + 2]; q.w = ew[4 * j + 3];
              s_q[slot * 4 + j] = q;
            }
          }
          q_len += n_new;
          q_back += n_newh;
          gf_wave_lds_sync();
        }
      }
      const unsigned int slot_b = gf_wave_append_lds(undecided, &s_cnt);
#ifdef GF_ENTRY_DIRECT_STORE
      constexpr bool direct_store = true;
#else
      constexpr bool direct_store = PW > 16;  // (the 20-word form has no registers to spare for the staging)
#endif
      if constexpr (direct_store) {
        if (undecided) {
          uint32_t pkw[PW + 1];
#pragma unroll
          for (int j = 0; j < PW; ++j) pkw[j] = gf_cut_pk(s_pk, w0, sh, j);
          pkw[PW] = 0;
          gf_entry_store<PW>(my_list + slot_b, (uint32_t)r, e_v1v2, e_todo, pkw);
        }
      } else {
      // The wave's entries are neighbours in the list: they go through the (now dead) LDS tile so
      // that a store instruction writes 1 KB of consecutive bytes instead of 16 bytes every 64
      // (a lane storing its own entry covers a quarter of each line per instruction).
      const uint64_t um = __ballot(undecided);
      if (um) {
        constexpr int EW = GfPipeEntryW<PW>::EW;
        constexpr int CH = PK_WORDS / (4 * EW);  // entries the tile holds
        const int cnt = __popcll(um), rank = gf_lanes_below(um);
        const unsigned int base =
            (unsigned int)__builtin_amdgcn_readlane((int)(slot_b - (unsigned int)rank), __builtin_ctzll(um));
        uint32_t ew[4 * EW];
#pragma unroll
        for (int j = 0; j < 4 * EW; ++j) ew[j] = 0;
        if (undecided) {
          ew[0] = (uint32_t)r;
          ew[1] = e_v1v2;
#pragma unroll
          for (int k = 0; k < NT; ++k) ew[2 + k] = e_todo[k];
#pragma unroll
          for (int j = 0; j < PW; ++j) ew[2 + NT + j] = gf_cut_pk(s_pk, w0, sh, j);
        }
        uint4* s_ent = (uint4*)s_pk;
        for (int c0 = 0; c0 < cnt; c0 += CH) {
          gf_wave_lds_sync();  // every lane has cut its words / the previous chunk has been read
          if (undecided && rank >= c0 && rank < c0 + CH) {
#pragma unroll
            for (int j = 0; j < EW; ++j)
              s_ent[(rank - c0) * EW + j] = make_uint4(ew[4 * j], ew[4 * j + 1], ew[4 * j + 2], ew[4 * j + 3]);
          }
          gf_wave_lds_sync();
          const int nvec = (cnt - c0 < CH ? cnt - c0 : CH) * EW;
          uint4* dst = (uint4*)(my_list + base + c0);
          for (int k = lane; k < nvec; k += 64) dst[k] = s_ent[k];
        }
      }
      }
      if constexpr (QUEUED) {
        while (q_len >= 64) q_len = gf_filter_queue_pass<PW>(T, s_q, q_len, lane, counts, my_list, &s_cnt);
        while (q_back >= 64) q_back = gf_high_queue_pass<PW>(T, s_q, q_back, lane, counts, my_list, &s_cnt);
      }
      if (batch_max > lmax) {  // (wave-uniform) batches with longer reads only
        const unsigned int s1 = gf_wave_append(long1k, ctr + 2);
        if (long1k) list_long[s1] = (uint32_t)r;
        const unsigned int s4 = gf_wave_append(long4k, ctr + 3);
        if (long4k) list_long[n - 1 - (int64_t)s4] = (uint32_t)r;
      }
      r0 += nfit;
    }
  }
  if constexpr (QUEUED) {  // what is left of the wave's queue
    while (q_len > 0) q_len = gf_filter_queue_pass<PW>(T, s_q, q_len, lane, counts, my_list, &s_cnt);
    while (q_back > 0) q_back = gf_high_queue_pass<PW>(T, s_q, q_back, lane, counts, my_list, &s_cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_cnt;
}

// ---- a block's chunk of entries through LDS (coalesced in, own entry out) ----
template <int PW>
__device__ __forceinline__ void gf_entries_to_lds(gf_u32x4* s_ent, const GfPipeEntryW<PW>* chunk, unsigned int cnt) {
  constexpr int EW = GfPipeEntryW<PW>::EW, EWP = EW + 1;
  const gf_u32x4* src = (const gf_u32x4*)chunk;
#pragma unroll
  for (int k = 0; k < EW; ++k) {
    const unsigned int g = (unsigned int)k * 256u + threadIdx.x;
    if (g < cnt * EW) s_ent[(g / EW) * EWP + g % EW] = src[g];
  }
}
template <int PW>
__device__ __forceinline__ void gf_entry_from_lds(const gf_u32x4* s_ent, uint32_t& r, uint32_t& v1v2,
                                                  uint32_t (&m)[GfPipeEntryW<PW>::NT], uint32_t (&pk)[PW + 1],
                                                  unsigned int e = 0xFFFFFFFFu) {
  constexpr int NT = GfPipeEntryW<PW>::NT, EW = GfPipeEntryW<PW>::EW, EWP = EW + 1;
  if (e == 0xFFFFFFFFu) e = threadIdx.x;  // (entry e of the chunk; by default the thread's own)
  uint32_t w[4 * EW];
#pragma unroll
  for (int j = 0; j < EW; ++j) {
    const gf_u32x4 q = s_ent[e * EWP + j];
    w[4 * j] = q.x; w[4 * j + 1] = q.y; w[4 * j + 2] = q.z; w[4 * j + 3] = q.w;
  }
  r = w[0];
  v1v2 = w[1];
#pragma unroll
  for (int k = 0; k < NT; ++k) m[k] = w[2 + k];
#pragma unroll
  for (int j = 0; j < PW; ++j) pk[j] = w[2 + NT + j];
  pk[PW] = 0;
}

// ---- K_filter / K_buckets: the undecided reads, in two dense passes ----
// K_filter (thread per undecided read) asks the presence filter about every unverified
// window, two windows per look-up (they share a 14-mer; L2 hits only): a window the filter
// rules out cannot vote, so with P windows left  v1 + P < 20 or v2 + P < 10  ->  [].  The
// reads that remain (the filter's false positives and the reads that really hit the
// table) are compacted in place to the front of the block's list region, their `todo`
// replaced by the windows still standing.
// K_buckets (thread per remaining read) probes those windows' buckets, as many at a time
// as must miss before the read can die, h = windows that voted; stops as soon as
// v1 + h + left < 20 or v2 + h + left < 10 -> [].  Survivors go to the exact kernel.
// (One kernel doing both kept every wave in the bucket loop for as long as its unluckiest
// lane: most of its instructions were executed for a handful of lanes.)
template <int PW>
__global__ __launch_bounds__(256) void gf_k_probe_filter(GfTable T, GfPipeEntryW<PW>* __restrict__ list_b,
                                                         const unsigned int* __restrict__ blk_cnt, int64_t per_block,
                                                         uint8_t* __restrict__ counts,
                                                         unsigned int* __restrict__ blk_cnt2, int phase, int nparts) {
  // nparts 1: the whole filter in one pass.  A filter larger than an XCD's L2 is asked in nparts
  // passes instead, each touching one part of its words (which then stays in the L2): phase p
  // asks the look-ups that fall in part p and leaves the others standing.
  //
  // Reads WITH a candidate diagonal (v1 > 0) ask every window the verification left open, part by part, and die
  // by  v1 + P < 20 or v2 + P < 10;  after the last part they are marked GF_ENTRY_FILTERED.
  // Reads WITHOUT one (background, mostly; they come here when the filter is too large for seed+verify to ask it
  // itself) go by the bound of gf_table.h (r03): part by part every pair of windows is asked, and after the last part
  // the bound decides — far stronger than "fewer than 20 windows left": the bucket pass behind sees half the reads.
  // (Asking only the even pairs first, as seed+verify does, needs a third launch over the survivors for a filter
  // in two parts, and a launch over the list costs 0.7 ms whatever it asks: 1.85 ms against 1.58, NOTEBOOK.md §5.)
  // With the whole filter in one part the even pairs go first and the bound may stop the read before the odd ones.
  // The block's 256 entries of a round go in and out through LDS: a lane fetching its own 64-byte entry as four
  // 16-byte loads touches a line per load (64 lanes, 64 lines, four times over), the block fetching the chunk as one
  // contiguous run touches each line once — a third of this kernel's L2 requests were its entries (r03, §5).
  // r04: the kernel is bound by instruction issue, and its lanes were busy 39 % of the time (SQ_THREAD_CYCLES_VALU):
  // an entry WITH a candidate diagonal takes another path (forty look-ups of its own) than one without, they are a per
  // cent or two of the list, and one of them among a wave's 64 entries makes the whole wave run both paths — 62 % of
  // the waves did.  So a chunk's entries are dealt out by kind: the ones without a candidate to the block's first
  // threads, the ones with to its last; at most one of the four waves holds both kinds.
  constexpr int EW = GfPipeEntryW<PW>::EW, EWP = EW + 1;  // (+1 vector of padding per entry: LDS banks)
  __shared__ gf_u32x4 s_ent[256 * EWP];
  __shared__ unsigned short s_perm[256];  // thread -> entry of the chunk it works on (0xFFFF: none)
  __shared__ unsigned int s_kind[2][4];   // entries without / with a candidate, per wave of the block
  __shared__ unsigned int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const unsigned int nb = blk_cnt[blockIdx.x];
  const uint32_t part_words = (T.bloom_words + (uint32_t)nparts - 1) / (uint32_t)nparts;
  const uint32_t part_lo = (uint32_t)phase * part_words, part_hi = part_lo + part_words;
  const bool last_part = phase == nparts - 1;
  GfPipeEntryW<PW>* my_list = list_b + (int64_t)blockIdx.x * per_block;
  for (unsigned int t0 = 0; t0 < nb; t0 += 256) {
    const unsigned int t = t0 + threadIdx.x;
    const unsigned int cnt = nb - t0 < 256u ? nb - t0 : 256u;
    gf_entries_to_lds<PW>(s_ent, my_list + t0, cnt);
    s_perm[threadIdx.x] = 0xFFFFu;
    __syncthreads();
    {  // deal the chunk's entries out by kind (see above)
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      bool with_cand = false;
      if (t < nb) {
        const gf_u32x4 q0 = s_ent[threadIdx.x * EWP];
        const uint32_t h = q0.y;  // v1 | v2 << 8 | flags
        with_cand = (h & GF_ENTRY_FILTERED) == 0 && T.bloom_words != 0 && !(PW == 10 && (h & 0xFFFFu) == 0);
      }
      const uint64_t mc = __ballot(with_cand), mn = __ballot(t < nb && !with_cand);
      if (lane == 0) {
        s_kind[0][wv] = (unsigned int)__popcll(mn);
        s_kind[1][wv] = (unsigned int)__popcll(mc);
      }
      __syncthreads();
      unsigned int before_n = 0, before_c = 0;
      for (int k = 0; k < wv; ++k) {
        before_n += s_kind[0][k];
        before_c += s_kind[1][k];
      }
      if (t < nb) {
        const unsigned int pos = with_cand ? 255u - (before_c + (unsigned int)gf_lanes_below(mc))
                                           : before_n + (unsigned int)gf_lanes_below(mn);
        s_perm[pos] = (unsigned short)threadIdx.x;
      }
      __syncthreads();
    }
    const unsigned int ent = s_perm[threadIdx.x];
    bool alive = false;
    constexpr int NT = GfPipeEntryW<PW>::NT;
    uint32_t r = 0, v1v2 = 0, m[NT], pk[PW + 1];
    uint32_t pp[NT];  // windows the filter could not rule out
#pragma unroll
    for (int k = 0; k < NT; ++k) m[k] = pp[k] = 0;
    if (ent != 0xFFFFu) {
      gf_entry_from_lds<PW>(s_ent, r, v1v2, m, pk, ent);
      const bool filtered = (v1v2 & GF_ENTRY_FILTERED) != 0;  // every window listed has been through the filter
      const int v1 = (int)(v1v2 & 0xFFu), v2 = (int)((v1v2 >> 8) & 0xFFu);
      const bool no_candidate = PW == 10 && !filtered && v1 == 0 && v2 == 0 && T.bloom_words != 0;
      bool dead = false;
      if (filtered) {
#pragma unroll
        for (int k = 0; k < NT; ++k) pp[k] = m[k];
      } else if (no_candidate) {
        // pairs of parity `par` whose filter word lies in this launch's part
        auto ask = [&](uint32_t par) {
#pragma unroll
          for (int j = 0; j < PW; ++j) {
            const uint32_t byte = (m[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            uint32_t word[2], bits[2];
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
              const uint32_t s14 = __builtin_amdgcn_alignbit(pk[j + 1], pk[j], 16u * (uint32_t)u2 + 8u * par + 4u) & 0x0FFFFFFFu;
              const uint32_t h2 = GF_BLOOM_HASH((s14));
              bits[u2] = GF_BLOOM_BITS(h2);
              const uint32_t widx = GF_BLOOM_WORD(h2, T.bloom_words);
              const bool mine = widx >= part_lo && widx < part_hi && (byte & (3u << (4 * u2 + 2 * par))) != 0;
              word[u2] = 0xFFFFFFFFu;  // (not this part's, or nobody to ask for: stands)
              if (mine) word[u2] = T.bloom[widx];
            }
            uint32_t fail2 = 0;
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) fail2 |= ((bits[u2] & ~word[u2]) != 0 ? 1u : 0u) << (4 * u2);
            fail2 <<= 2 * par;
            m[j >> 2] &= ~((fail2 | (fail2 << 1)) << (8 * (j & 3)));
          }
        };
        auto bound = [&]() {
          uint32_t x[NT];
#pragma unroll
          for (int k = 0; k < NT; ++k) x[k] = (m[k] | (m[k] >> 1)) & 0x55555555u;
          return gf_vote_bound_pairs<4 * PW - 3>(x);
        };
        if (nparts == 1) {   // the whole filter at hand: even pairs, the bound, and only then the odd ones
          ask(0u);
          dead = bound() < GF_MAJOR_KEYS / 2;
          if (!dead) {
            ask(1u);
            dead = bound() < GF_MAJOR_KEYS / 2;
          }
          v1v2 |= GF_ENTRY_FILTERED;
        } else if (!last_part) {  // part by part: every pair of this part; nothing can be decided yet
          ask(0u);
          ask(1u);
        } else {                  // the last part: its even pairs may already settle it (pairs not asked stand)
          ask(0u);
          dead = bound() < GF_MAJOR_KEYS / 2;
          if (!dead) {
            ask(1u);
            dead = bound() < GF_MAJOR_KEYS / 2;
          }
          v1v2 |= GF_ENTRY_FILTERED;
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) pp[k] = m[k];
      } else if (T.bloom_words) {
        int npos = 0, rem = 0;  // not ruled out / not asked yet
#pragma unroll
        for (int k = 0; k < NT; ++k) rem += __popc(m[k]);
        // windows 2q and 2q+1 share the 14-mer at bases 4q+2 .. 4q+15: one lookup for both.
        // Fully unrolled over the pairs (compile-time shifts on the words in registers),
        // four look-ups in flight per step.
#pragma unroll
        for (int q0 = 0; q0 < 4 * PW; q0 += 4) {
          if (!dead) {
            uint32_t word[4], bits[4], both[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = q0 + u;
              const int wbit = (2 * q) & 31, wword = (2 * q) >> 5;
              both[u] = wword < NT ? (m[wword] >> wbit) & 3u : 0u;
              const int b0 = 4 * q + 2;  // first base of the shared 14-mer
              const int j = b0 >> 4;
              const uint32_t sh14 = 2u * (uint32_t)(b0 & 15);
              const uint32_t s14 = __builtin_amdgcn_alignbit(pk[j + 1], pk[j], sh14) & 0x0FFFFFFFu;
              const uint32_t h2 = GF_BLOOM_HASH((s14));
              bits[u] = GF_BLOOM_BITS(h2);
              const uint32_t widx = GF_BLOOM_WORD(h2, T.bloom_words);
              // (a look-up outside this phase's half stands as if the filter had let it pass)
              const bool mine = widx >= part_lo && widx < part_hi;
              word[u] = mine ? 0u : bits[u];
              if (both[u] && mine) word[u] = T.bloom[widx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int q = q0 + u;
              const int cnt = (int)(both[u] & 1u) + (int)(both[u] >> 1);
              rem -= cnt;
              if (both[u] && (word[u] & bits[u]) == bits[u]) {
                pp[(2 * q) >> 5 < NT ? (2 * q) >> 5 : 0] |= both[u] << ((2 * q) & 31);
                npos += cnt;
              }
            }
            // even if every window not asked yet could vote, the gate is out of reach
            dead = (v1 + npos + rem < GF_MAJOR_KEYS / 2) || (v2 + npos + rem < GF_MINOR_KEYS / 2);
          }
        }
        if (last_part) v1v2 |= GF_ENTRY_FILTERED;  // every part has been asked: what stands has passed the filter
      } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) pp[k] = m[k];
        int left = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) left += __popc(m[k]);
        dead = !filtered && ((v1 + left < GF_MAJOR_KEYS / 2) || (v2 + left < GF_MINOR_KEYS / 2));
      }
      if (dead) counts[r] = 0;
      alive = !dead;
    }
    const unsigned int base = s_cnt;  // (stable: the last appends were before the barrier above)
    __syncthreads();  // this chunk's entries are in registers: the staging area and the list may be overwritten
    const unsigned int slot = gf_wave_append_lds(alive, &s_cnt);
    if (alive) {
      uint32_t w[4 * EW];
      w[0] = r;
      w[1] = v1v2;
#pragma unroll
      for (int k = 0; k < NT; ++k) w[2 + k] = pp[k];
#pragma unroll
      for (int j = 0; j < PW; ++j) w[2 + NT + j] = pk[j];
#pragma unroll
      for (int j = 2 + NT + PW; j < 4 * EW; ++j) w[j] = 0;
#pragma unroll
      for (int j = 0; j < EW; ++j) {
        gf_u32x4 q;
        q.x = w[4 * j]; q.y = w[4 * j + 1]; q.z = w[4 * j + 2]; q.w = w[4 * j + 3];
        s_ent[(slot - base) * EWP + j] = q;
      }
    }
    __syncthreads();
    {  // the survivors of the chunk, compacted, as one contiguous run behind the earlier ones (never beyond t0 + cnt)
      const unsigned int n_alive = s_cnt - base;
      gf_u32x4* dst = (gf_u32x4*)(my_list + base);
#pragma unroll
      for (int k = 0; k < EW; ++k) {
        const unsigned int g = (unsigned int)k * 256u + threadIdx.x;
        if (g < n_alive * EW) dst[g] = s_ent[(g / EW) * EWP + g % EW];
      }
    }
    __syncthreads();  // the next chunk's loads overwrite the staging area
  }
  if (threadIdx.x == 0) blk_cnt2[blockIdx.x] = s_cnt;
}

// pass B of seed+verify on an entry's words (registers, the read starts at bit 0 of pk[0]): the windows whose 16 bases
// equal the bases of site K + 2w and whose site is the only one of its key, one bit per stride-2 window.  No
// "bad base" stream here: the caller intersects the result with windows known to be clean.
// HIGHBIT: the windows whose 16 bases equal the bases of site K + 2w and whose site belongs to a key of six sites or
// more (the odd flag bits, gf_k_index_side) instead: windows PROVEN unable to vote.
template <int PW, bool HIGHBIT = false>
__device__ __forceinline__ void gf_verify_words(const GfTable& T, const uint32_t (&pk)[PW + 1], uint32_t K,
                                                uint32_t (&vmb)[GfPipeEntryW<PW>::NT]) {
  constexpr int NT = GfPipeEntryW<PW>::NT;
#pragma unroll
  for (int k = 0; k < NT; ++k) vmb[k] = 0;
  const uint2* gp = (const uint2*)T.gdu + (K >> 4);
  const uint32_t bo = 2u * (K & 15u);
  uint2 gw[PW + 1];
#pragma unroll
  for (int j = 0; j < PW + 1; ++j) gw[j] = gp[j];
  uint32_t zz_cur;
  {
    const uint32_t x = pk[0] ^ __builtin_amdgcn_alignbit(gw[1].x, gw[0].x, bo);
    zz_cur = (x | (x >> 1)) & 0x55555555u;
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    uint32_t zz_next = 0x55555555u;
    if (j + 1 < PW) {
      const uint32_t x = pk[j + 1] ^ __builtin_amdgcn_alignbit(gw[j + 2 <= PW ? j + 2 : PW].x, gw[j + 1].x, bo);
      zz_next = (x | (x >> 1)) & 0x55555555u;
    }
    uint32_t u = __builtin_amdgcn_alignbit(gw[j + 1].y, gw[j].y, bo);
    if (HIGHBIT) u >>= 1;
    const uint32_t ver = gf_clean16(zz_cur, zz_next) & u & 0x11111111u;
    vmb[j >> 2] |= gf_gather_nibble_lsb(ver) << (8 * (j & 3));
    zz_cur = zz_next;
  }
}

template <int PW>
__global__ __launch_bounds__(256) void gf_k_probe_buckets(GfTable T, const GfPipeEntryW<PW>* __restrict__ list_b,
                                                          const unsigned int* __restrict__ blk_cnt2,
                                                          int64_t per_block, uint8_t* __restrict__ counts,
                                                          uint32_t* __restrict__ list_c,
                                                          unsigned int* __restrict__ ctr) {
  // the read's codes live in LDS for the duration of its probes ([word][thread]: each
  // thread reads only its own column, conflict-free)
  __shared__ uint32_t s_pk[(PW + 1) * 256];
  const unsigned int nb = blk_cnt2[blockIdx.x];
  const GfPipeEntryW<PW>* my_list = list_b + (int64_t)blockIdx.x * per_block;
  const unsigned int nb_round = (nb + 63u) & ~63u;  // whole waves stay in the loop for the ballot
  // (no block barrier in this loop: a wave's probes take as long as its unluckiest lane, and fetching the entries
  //  block-wide through LDS, as gf_k_probe_filter does, made every wave wait for the block's — 0.15 -> 0.29 ms, r03)
  for (unsigned int t = threadIdx.x; t < nb_round; t += blockDim.x) {
    bool to_full = false;
    uint32_t r = 0;
    if (t < nb) {
      constexpr int NT = GfPipeEntryW<PW>::NT;
      uint32_t v1v2, p[NT], pk[PW + 1];
      gf_entry_load<PW>(my_list + t, r, v1v2, p, pk);
      int v1 = (int)(v1v2 & 0xFFu);
      const int v2 = (int)((v1v2 >> 8) & 0xFFu);
#pragma unroll
      for (int j = 0; j <= PW; ++j) s_pk[j * 256 + threadIdx.x] = pk[j];  // the loop indexes the words dynamically
      int left = 0;
#pragma unroll
      for (int k = 0; k < NT; ++k) left += __popc(p[k]);
      int h = 0;
      // A read that comes without a candidate diagonal (seed+verify found none: it asks only its first seed when the
      // filter lives beyond the L2, r03) gets one here: the first window whose bucket holds a UNIQUE site names K, the
      // read is verified against the genes on K like in seed+verify, and the windows that match there — they vote for
      // K and for nothing else — leave the list: an on-target read costs two or three probes instead of sixty.
      bool want_k = v1 == 0 && v2 == 0;
      uint32_t p0[NT], voted[NT];  // the windows standing on entry (all clean); those probed so far that voted
#pragma unroll
      for (int k = 0; k < NT; ++k) { p0[k] = p[k]; voted[k] = 0; }
      bool dead = (v1 + left < GF_MAJOR_KEYS / 2) || (v2 + left < GF_MINOR_KEYS / 2);
      while (!dead && left > 0) {
        // the read dies only after at least `need` more probes miss: issue that many (up to
        // 4) bucket probes together instead of one round trip each
        const int needA = v1 + h + left - (GF_MAJOR_KEYS / 2 - 1);
        const int needB = v2 + h + left - (GF_MINOR_KEYS / 2 - 1);
        int need = needA < needB ? needA : needB;
        need = need < 1 ? 1 : (need > 4 ? 4 : need);
        uint32_t key[4];
        bool act[4];
        int wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint32_t any = 0;
#pragma unroll
          for (int k = 0; k < NT; ++k) any |= p[k];
          act[u] = u < need && any;
          int w = 0;
          if (act[u]) {  // lowest window still standing
            bool taken = false;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              if (!taken && p[k]) {
                w = 32 * k + __builtin_ctz(p[k]);
                p[k] &= p[k] - 1;
                taken = true;
              }
            }
          }
          wv[u] = w;
          const int j = w >> 3;
          const uint32_t sh = 4u * (uint32_t)(w & 7);
          key[u] = __builtin_amdgcn_alignbit(s_pk[(j + 1) * 256 + threadIdx.x], s_pk[j * 256 + threadIdx.x], sh);
        }
        uint32_t val[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (act[u]) val[u] = gf_lookup<GF_PROBE_NT>(T, key[u]);
        uint32_t K = GF_NONE_LIN;
        uint32_t Kh = GF_NONE_LIN;  // a copy of the repeat this read lies in: the diagonal of a HIGH key's representative site
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (act[u]) {
            const uint32_t ty = val[u] >> GF_TYPE_SHIFT;
            if (ty == GF_TYPE_UNIQUE || ty == GF_TYPE_DUPES) {
              h += 1;
              voted[wv[u] >> 5] |= 1u << (wv[u] & 31);
              if (want_k && ty == GF_TYPE_UNIQUE && K == GF_NONE_LIN) K = (val[u] & GF_LIN_MASK) - 2u * (uint32_t)wv[u];
            } else if (ty == GF_TYPE_HIGH && (val[u] & GF_LIN_MASK) != GF_LIN_MASK && Kh == GF_NONE_LIN) {
              Kh = (val[u] & GF_LIN_MASK) - 2u * (uint32_t)wv[u];
            }
            left -= 1;
          }
        }
        if (PW == 10 && Kh != GF_NONE_LIN) {
          // r04: windows whose bases equal a site of a key with six sites or more are proven unable to vote — all of
          // them at once, where each used to cost a probe of its own (a read inside a repeat: fifty probes in a row)
          uint32_t hk[NT];
          gf_verify_words<PW, true>(T, pk, Kh, hk);
          left = 0;
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            p[k] &= ~hk[k];  // (windows past the read's end compared garbage: none of them stands in p)
            left += __popc(p[k]);
          }
        }
        if (K != GF_NONE_LIN) {
          want_k = false;
          uint32_t vk[NT];
          gf_verify_words<PW>(T, pk, K, vk);
          int nv = 0, nh = 0;
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            vk[k] &= p0[k];                 // clean windows only (the words past the read's end compared garbage)
            nv += __popc(vk[k]);
            nh += __popc(vk[k] & voted[k]);  // probed already and counted in h: they are v1's now
            p[k] &= ~vk[k];
          }
          v1 = nv;
          h -= nh;
          left = 0;
#pragma unroll
          for (int k = 0; k < NT; ++k) left += __popc(p[k]);
        }
        dead = (v1 + h + left < GF_MAJOR_KEYS / 2) || (v2 + h + left < GF_MINOR_KEYS / 2);
      }
      if (dead) counts[r] = 0;
      else to_full = true;
    }
    const unsigned int slot = gf_wave_append(to_full, ctr + 1);
    if (to_full) list_c[slot] = r;
  }
}

// ---- K_full: the exact wave-per-read kernel over a list of read indices ----
template <int LCAP, int WAVES, bool PACKED = false>
__global__ __launch_bounds__(WAVES * 64, LCAP <= 256 ? 8 : (LCAP <= 1024 ? 3 : 1)) void gf_k_map_reads_list(
    GfTable T, const uint8_t* __restrict__ bases, const uint32_t* __restrict__ g_pk,
    const uint16_t* __restrict__ g_iv, const int64_t* __restrict__ offsets,
    const uint32_t* __restrict__ list, int64_t stride, const unsigned int* __restrict__ n_list,
    uint8_t* __restrict__ counts, gf_seqmatch* __restrict__ matches) {
  __shared__ GfMapSmem<LCAP> smem[WAVES];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  GfMapSmem<LCAP>& S = smem[wib];
  const unsigned int nl = *n_list;
  for (unsigned int k = blockIdx.x * WAVES + wib; k < nl; k += gridDim.x * WAVES) {
    const int64_t r = list ? (int64_t)list[(int64_t)k * stride] : (int64_t)k;  // (no list: reads 0 .. *n_list - 1)
    const int64_t off0 = T.fixed_len ? r * (int64_t)T.fixed_len : offsets[r];
    const int L = T.fixed_len ? T.fixed_len : (int)(offsets[r + 1] - off0);
    gf_wave_lds_sync();
    uint32_t sh;
    if constexpr (PACKED) sh = gf_stage_read_packed<LCAP>(S, g_pk, g_iv, off0, L, lane);
    else sh = gf_stage_read<LCAP>(S, bases + off0, L, lane);
    gf_wave_lds_sync();
    // reads of up to 256 bases: seeds + verification against the genes, probes only for the windows
    // neither explains (the vote list is the same as probing every window; gf_map_kernels.h, producer B)
    int nvotes;
    if constexpr (LCAP <= 256) nvotes = gf_first_pass_seed_verify<LCAP>(T, S, L, sh, lane);
    else nvotes = gf_first_pass_probe_all<LCAP>(T, S, L, sh, lane);
    gf_finish_read<LCAP>(T, S, L, sh, nvotes, lane, r, counts, matches);
  }
}

// ---- ASCII bases -> the packed form the PACKED kernels take: thread per 16-base chunk ----
// pk[c] = 2-bit codes of bases 16c .. 16c+15 (base j in bits 2j, 2j+1; gf_table.h), iv[c] bit j = that
// base is not one of A C G T (bases at or beyond n_bases count as such).
__global__ __launch_bounds__(256) void gf_k_pack_bases(const uint8_t* __restrict__ bases, int64_t n_bases,
                                                       uint32_t* __restrict__ pk, uint16_t* __restrict__ iv,
                                                       int64_t n_chunks) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b0 = 16 * c;
    uint4 q = make_uint4(0, 0, 0, 0);
    uint32_t tail_bad = 0;
    if (b0 + 16 <= n_bases) {
      const gf_u32x4 t = __builtin_nontemporal_load((const gf_u32x4*)(bases + b0));  // (any byte alignment)
      q = make_uint4(t.x, t.y, t.z, t.w);
    } else {
      uint32_t w[4] = {0, 0, 0, 0};
      for (int j = 0; j < 16; ++j) {
        if (b0 + j < n_bases) w[j >> 2] |= (uint32_t)bases[b0 + j] << (8 * (j & 3));
        else tail_bad |= 1u << j;
      }
      q = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint32_t code32, bad16;
    gf_convert16(q, code32, bad16);
    pk[c] = code32;
    iv[c] = (uint16_t)(bad16 | tail_bad);
  }
}
