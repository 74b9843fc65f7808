// K2/K3 — Indexer::map_read on the device, one wavefront (64 lanes) per read.
//
// Restates src/core/indexer.rs:252-538 (map_read) and :616-679 (segment_mask):
//   first pass   every second 16-base window votes for the diagonal
//                GenePos{contig, position - i} of each stored site (:275-321)
//   top two      by (count desc, key64 asc), key 0 skipped (:323-346)
//   gate         2*count1 >= 40 and 2*count2 >= 20 (:353-360)
//   second pass  every window classified TOP / SECOND / NONE against the two
//                diagonals (+-1 in key64 space), spread over 16 bases (:362-521)
//   mismatch gate (:523-535) and segment_mask (:616-679)
//
// Wave-per-read keeps every branch of the rare path (survivors of the gate)
// wave-uniform.  Per read the wave
//   1. loads the read's bytes with coalesced dword loads, converts them once to a
//      2-bit stream + invalid-bit stream in LDS;
//   2. produces the first-pass vote list in LDS (two interchangeable producers,
//      see "first pass" below);
//   3. counts equal diagonals by repeated ballot ("peel"), ranks the top two;
//   4. survivors only: second pass, mask, segment_mask.
// Votes are 32-bit site codes (gf_table.h); only diagonals with >= 10 votes are
// decoded to (contig, position) — nothing smaller can survive the gate.
//
// First pass, producer A ("probe all", any read length): lane = window, one
// 64-byte bucket probe of the HBM/Infinity-Cache resident table per window.
// The chip serves ~56 G such L2-missing requests per second whatever their size
// (tools/mb_gather.hip), which is what bounds this producer.
//
// First pass, producer B ("seed + verify", reads up to 256 bases): exact, but
// spends far fewer L2-missing requests:
//   * 4 seed windows are probed; every UNIQUE seed hit names a candidate diagonal;
//   * each candidate is verified against the genes themselves, stored for both
//     strands in site-code space (gf_table.h: gdu): window i is *verified* when its
//     16 bases equal the bases of site K+i and that site is flagged as the only site
//     of its key — then the table would return exactly that one site, i.e. exactly one
//     vote for the candidate, so the probe is skipped.  No contig decode, no strand
//     logic; one diagonal costs ~3 cache lines instead of 68;
//   * bound check: a diagonal can collect at most one vote per window, so with U
//     windows still unknown the best two counts are at most v1+U and v2+U; if
//     v1+U < 20 or v2+U < 10 the read cannot pass the gate -> [] without probing;
//   * otherwise the unknown windows are probed like producer A, all but the last 19
//     first: when no window has produced a vote by then, count1 <= 19 -> [].
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gfmatch.h"
#include "gf_table.h"

#define GF_NONE_LIN 0xFFFFFFFFu

__device__ __forceinline__ void gf_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int gf_lanes_below(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

typedef uint32_t gf_u32x4 __attribute__((ext_vector_type(4)));

// One probe: the 64-byte bucket of `key`.  Returns the slot's val with the
// overflow bit cleared, 0 when the key is absent.  NT = load the bucket with the
// non-temporal hint (a line that is used once should not push the L2-resident presence
// filter out).
template <bool NT = false>
__device__ __forceinline__ uint32_t gf_lookup(const GfTable& T, uint32_t key) {
  uint32_t b = gf_bucket_of(key, T.nbuckets);
  for (uint32_t guard = 0; guard <= T.nbuckets; ++guard) {
    const uint4* p = (const uint4*)(T.slots + (size_t)b * GF_SLOTS_PER_BUCKET);
    uint4 q0, q1, q2, q3;
    if (NT) {
      const gf_u32x4* pv = (const gf_u32x4*)p;
      const gf_u32x4 a0 = __builtin_nontemporal_load(pv), a1 = __builtin_nontemporal_load(pv + 1),
                     a2 = __builtin_nontemporal_load(pv + 2), a3 = __builtin_nontemporal_load(pv + 3);
      q0 = make_uint4(a0.x, a0.y, a0.z, a0.w); q1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
      q2 = make_uint4(a2.x, a2.y, a2.z, a2.w); q3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
    } else {
      q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
    }
    uint32_t r = 0;
    // slot = (key << 32) | val, little endian: .x/.z = val, .y/.w = key
    r = (q0.y == key && (q0.x & GF_VAL_LOW)) ? q0.x : r;
    r = (q0.w == key && (q0.z & GF_VAL_LOW)) ? q0.z : r;
    r = (q1.y == key && (q1.x & GF_VAL_LOW)) ? q1.x : r;
    r = (q1.w == key && (q1.z & GF_VAL_LOW)) ? q1.z : r;
    r = (q2.y == key && (q2.x & GF_VAL_LOW)) ? q2.x : r;
    r = (q2.w == key && (q2.z & GF_VAL_LOW)) ? q2.z : r;
    r = (q3.y == key && (q3.x & GF_VAL_LOW)) ? q3.x : r;
    r = (q3.w == key && (q3.z & GF_VAL_LOW)) ? q3.z : r;
    if (r) return r & GF_VAL_LOW;
    if (!(q0.x & GF_VAL_OVF)) return 0;
    b = (b + 1 == T.nbuckets) ? 0 : b + 1;
  }
  return 0;
}

// Sites of a window as site codes shifted to the window's diagonal (lin - i).
__device__ __forceinline__ int gf_sites(const GfTable& T, uint32_t val, uint32_t i, uint32_t v[5]) {
  uint32_t type = val >> GF_TYPE_SHIFT;
  if (type == GF_TYPE_UNIQUE) {
    v[0] = (val & GF_LIN_MASK) - i;
    return 1;
  }
  if (type == GF_TYPE_DUPES) {
    int cnt = (int)((val >> GF_DUPE_COUNT_SHIFT) & 7u);
    const uint32_t* d = T.dupes + (val & GF_DUPE_START_MASK);
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (k < cnt) ? d[k] - i : 0u;
    return cnt;
  }
  return 0;  // absent or HIGH
}

// contig owning site code `lin` (wave-uniform callers): smallest c with lin < lin_hi[c]
__device__ __forceinline__ int gf_contig_of(const GfTable& T, uint32_t lin) {
  int lo = 0, hi = T.n_genes - 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (lin < T.lin_hi[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// the same for a wave-uniform `lin`, without the chain of dependent loads: lin_hi is
// ascending, so the contig is the number of genes whose interval ends at or below lin;
// 64 genes per coalesced load, all loads independent
__device__ __forceinline__ int gf_contig_of_wave(const GfTable& T, uint32_t lin, int lane) {
  int c = 0;
  for (int base = 0; base < T.n_genes; base += 64) {
    const int g = base + lane;
    const uint32_t hi = g < T.n_genes ? T.lin_hi[g] : 0xFFFFFFFFu;
    c += __popcll(__ballot(lin >= hi));
  }
  return c < T.n_genes ? c : T.n_genes - 1;
}

// key64 (indexer.rs:698-706) -> site code, GF_NONE_LIN when no vote can carry it.
// The two table loads are unconditional (clamped index) so that several calls overlap.
__device__ __forceinline__ uint32_t gf_lin_of_key(const GfTable& T, int64_t key) {
  const int64_t c = key >> 32;
  const bool in_range = c >= 0 && c < (int64_t)T.n_genes;
  const int cc = in_range ? (int)c : 0;
  const int32_t d = (int32_t)(uint32_t)key;
  const int64_t len = T.n_genes > 0 ? (int64_t)T.gene_len[cc] : 0;
  const uint32_t base = T.n_genes > 0 ? T.lin_base[cc] : 0u;
  const bool ok = in_range && (int64_t)d < len && (int64_t)d >= -(len + (int64_t)GF_LIN_PAD);
  return ok ? base + (uint32_t)d : GF_NONE_LIN;
}

// inclusive end of the run starting at s (indexer.rs:644-661)
__device__ __forceinline__ int gf_run_end(const uint8_t* mask, int L, int s, int target) {
  int end = s + 1, g = 0;
  while (g < GF_ALLOWED_GAP && end + g < L) {
    int m = mask[end + g];
    if (m > target) break;
    if (m == target) {
      end += g + 1;
      g = 0;
      continue;
    }
    g += 1;
  }
  return end - 1;
}

template <int LCAP>
struct GfMapSmem {
  static constexpr int NWIN1 = (LCAP - GF_KMER) / 2 + 1;
  static constexpr int NVOTES = NWIN1 * 5;
  uint32_t codes[LCAP / 16 + 4];
  uint32_t inv[LCAP / 32 + 4];
  union {
    uint32_t votes[NVOTES];
    struct {
      uint8_t wcls[LCAP];
      uint8_t mask[LCAP];
    } p2;
  } u;
};

// ---- 1'. the same from dwords already in registers (software-prefetched by the caller):
// lane t holds aligned dwords t and t+64 of the read ----
template <int LCAP>
__device__ __forceinline__ void gf_stage_regs(GfMapSmem<LCAP>& S, uint32_t x0, uint32_t x1, int ndw, int lane) {
  uint8_t* codes_b = (uint8_t*)S.codes;
  uint8_t* inv_b = (uint8_t*)S.inv;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && ndw <= 64) break;
    const int t = lane + 64 * h;
    uint32_t code8, inv4;
    gf_convert4(h ? x1 : x0, code8, inv4);
    uint32_t nb = (uint32_t)__shfl_down((int)inv4, 1);
    if (t < ndw) {
      codes_b[t] = (uint8_t)code8;
      if (!(lane & 1)) inv_b[t >> 1] = (uint8_t)(inv4 | (nb << 4));
    }
  }
}

// ---- 1. read -> 2-bit stream + invalid bits in LDS; returns the byte misalignment ----
template <int LCAP>
__device__ __forceinline__ uint32_t gf_stage_read(GfMapSmem<LCAP>& S, const uint8_t* p, int L, int lane) {
  uint8_t* codes_b = (uint8_t*)S.codes;
  uint8_t* inv_b = (uint8_t*)S.inv;
  const uintptr_t addr = (uintptr_t)p;
  const uint32_t sh = (uint32_t)(addr & 3u);
  const uint32_t* pw = (const uint32_t*)(addr - sh);
  const int ndw = (int)((sh + (uint32_t)L + 3u) >> 2);
  for (int t0 = 0; t0 < ndw; t0 += 64) {
    int t = t0 + lane;
    uint32_t x = (t < ndw) ? pw[t] : 0u;
    uint32_t code8, inv4;
    gf_convert4(x, code8, inv4);
    uint32_t nb = (uint32_t)__shfl_down((int)inv4, 1);
    if (t < ndw) {
      codes_b[t] = (uint8_t)code8;
      if (!(lane & 1)) inv_b[t >> 1] = (uint8_t)(inv4 | (nb << 4));
    }
  }
  return sh;
}

// ---- 1p. the same from the packed form of the batch (gf_pack_bases_device: word c of g_pk / g_iv =
// bases 16c .. 16c+15 of the whole stream): a copy; returns the read's phase inside its first word ----
template <int LCAP>
__device__ __forceinline__ uint32_t gf_stage_read_packed(GfMapSmem<LCAP>& S, const uint32_t* __restrict__ g_pk,
                                                         const uint16_t* __restrict__ g_iv, int64_t off0, int L,
                                                         int lane) {
  const uint32_t sh = (uint32_t)(off0 & 15);
  const int64_t c0 = off0 >> 4;
  const int nw = (int)((sh + (uint32_t)L + 15u) >> 4);
  uint16_t* inv_h = (uint16_t*)S.inv;
  for (int t = lane; t < nw; t += 64) {
    S.codes[t] = g_pk[c0 + t];
    inv_h[t] = g_iv[c0 + t];
  }
  return sh;
}

// append up to 5 votes per lane to the LDS list (ballot + mbcnt prefix sums)
template <int LCAP>
__device__ __forceinline__ int gf_append_votes(GfMapSmem<LCAP>& S, int nvotes, int nv, const uint32_t v[5]) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const bool has = k < nv;
    const uint64_t m = __ballot(has);
    if (m == 0) break;
    if (has) S.u.votes[nvotes + gf_lanes_below(m)] = v[k];
    nvotes += __popcll(m);
  }
  return nvotes;
}

// ---- 2A. first pass, producer A: probe every stride-2 window ----
template <int LCAP>
__device__ __forceinline__ int gf_first_pass_probe_all(const GfTable& T, GfMapSmem<LCAP>& S, int L, uint32_t sh,
                                                       int lane) {
  const int nwin = ((L - GF_KMER) >> 1) + 1;
  int nvotes = 0;
  for (int w0 = 0; w0 < nwin; w0 += 64) {
    const int w = w0 + lane;
    const bool act = w < nwin;
    const uint32_t i = act ? 2u * (uint32_t)w : 0u;
    const uint32_t g = sh + i;
    const uint32_t key = gf_window(S.codes[g >> 4], S.codes[(g >> 4) + 1], g);
    const uint32_t bad = gf_flags16(S.inv[g >> 5], S.inv[(g >> 5) + 1], g);
    uint32_t val = 0;
    if (act && !bad) val = gf_lookup(T, key);
    uint32_t v[5];
    const int nv = gf_sites(T, val, i, v);
    nvotes = gf_append_votes(S, nvotes, nv, v);
  }
  return nvotes;
}

// ---- 2B. first pass, producer B: seed + verify (reads up to 256 bases = 2 windows per lane) ----
// Returns the number of votes in the LDS list, or -1 when the read provably fails the gate.
#define GF_ST_NONE 0u      /* window absent or invalid: never votes */
#define GF_ST_UNKNOWN 1u   /* not probed yet */
#define GF_ST_PROBED 2u    /* table result in val */
#define GF_ST_VERIFIED 3u  /* exactly one vote, for the diagonal in val */

template <int LCAP>
__device__ __forceinline__ int gf_first_pass_seed_verify(const GfTable& T, GfMapSmem<LCAP>& S, int L,
                                                         uint32_t sh, int lane) {
  static_assert(LCAP <= 256 + 14, "two stride-2 windows per lane");
  const int nwin = ((L - GF_KMER) >> 1) + 1;  // <= 128
  uint32_t key[2], val[2], st[2], wi[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int w = lane + 64 * h;
    const bool act = w < nwin;
    wi[h] = act ? 2u * (uint32_t)w : 0u;
    const uint32_t g = sh + wi[h];
    key[h] = gf_window(S.codes[g >> 4], S.codes[(g >> 4) + 1], g);
    const uint32_t bad = gf_flags16(S.inv[g >> 5], S.inv[(g >> 5) + 1], g);
    st[h] = (act && !bad) ? GF_ST_UNKNOWN : GF_ST_NONE;
    val[h] = 0;
  }

#if defined(GF_ABLATE) && GF_ABLATE <= 2
  return ((key[0] ^ key[1] ^ st[0] ^ (st[1] << 3)) == 0x12345677u) ? 1000 : -1;  // timing-only build: stop after key extraction
#endif
  // seeds: windows 0, 16, 32, 48 (all inside the first probing phase)
  const bool seed = (lane & 15) == 0 && st[0] == GF_ST_UNKNOWN;
  if (seed) {
    val[0] = gf_lookup(T, key[0]);
    st[0] = GF_ST_PROBED;
  }
#if defined(GF_ABLATE) && GF_ABLATE <= 3
  return ((key[0] ^ key[1] ^ st[0] ^ (st[1] << 3) ^ val[0]) == 0x12345677u) ? 1000 : -1;  // timing-only: stop after the seed probes
#endif
  int v1 = 0, v2 = 0;  // best two candidate counts (upper bounds of their final counts minus U)
  {
    const bool uniq = seed && (val[0] >> GF_TYPE_SHIFT) == GF_TYPE_UNIQUE;
    uint32_t myK = uniq ? (val[0] & GF_LIN_MASK) - wi[0] : GF_NONE_LIN;
    uint64_t todo = __ballot(uniq);
    while (todo != 0) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t K = (uint32_t)__builtin_amdgcn_readlane((int)myK, leader);
      const uint64_t same = __ballot(myK == K);
      todo &= ~same;
      int cnt = __popcll(same);  // seed windows that already voted for K
      // window i lies on candidate K iff its 16 bases equal the site K+i's bases in
      // site-code space and that site is the unique site of its key (gf_table.h)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bool ver = false;
        if (st[h] == GF_ST_UNKNOWN) {
          const uint32_t a = K + wi[h];
          const uint32_t gk = gf_window(T.gdu[2 * (a >> 4)], T.gdu[2 * (a >> 4) + 2], a);
          const uint32_t ubit = (T.gdu[2 * (a >> 4) + 1] >> (2u * (a & 15u))) & 1u;
          ver = ubit && gk == key[h];
        }
        if (ver) {
          st[h] = GF_ST_VERIFIED;
          val[h] = K;
        }
        cnt += __popcll(__ballot(ver));
      }
      if (cnt > v1) { v2 = v1; v1 = cnt; } else if (cnt > v2) { v2 = cnt; }
    }
  }

  // bound check: every other diagonal gets at most one vote per window that can still vote
  const bool can0 = st[0] == GF_ST_UNKNOWN || (st[0] == GF_ST_PROBED && (val[0] >> GF_TYPE_SHIFT) == GF_TYPE_DUPES);
  const bool can1 = st[1] == GF_ST_UNKNOWN;
  const int open_votes = __popcll(__ballot(can0)) + __popcll(__ballot(can1));
  if (v1 + open_votes < GF_MAJOR_KEYS / 2 || v2 + open_votes < GF_MINOR_KEYS / 2) return -1;
#if defined(GF_ABLATE) && GF_ABLATE <= 4
  return ((key[0] ^ key[1] ^ st[0] ^ (st[1] << 3) ^ val[0] ^ val[1] ^ (uint32_t)v1) == 0x12345677u) ? 1000 : -1;  // timing-only
#endif

  // probe what is still unknown: everything but the last 19 windows first
  const int tail0 = nwin - (GF_MAJOR_KEYS / 2 - 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (st[h] == GF_ST_UNKNOWN && lane + 64 * h < tail0) {
      val[h] = gf_lookup(T, key[h]);
      st[h] = GF_ST_PROBED;
    }
  }
  {
    int voted = 0, unknown = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t ty = val[h] >> GF_TYPE_SHIFT;
      const bool has = st[h] == GF_ST_VERIFIED ||
                       (st[h] == GF_ST_PROBED && (ty == GF_TYPE_UNIQUE || ty == GF_TYPE_DUPES));
      voted += __popcll(__ballot(has));
      unknown += __popcll(__ballot(st[h] == GF_ST_UNKNOWN));
    }
    // count1 <= number of windows that vote at all
    if (voted + unknown < GF_MAJOR_KEYS / 2) return -1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (st[h] == GF_ST_UNKNOWN) {
      val[h] = gf_lookup(T, key[h]);
      st[h] = GF_ST_PROBED;
    }
  }

  // vote list, identical in content to producer A's
  int nvotes = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t v[5];
    int nv = 0;
    if (st[h] == GF_ST_VERIFIED) {
      v[0] = val[h];
      v[1] = v[2] = v[3] = v[4] = 0;
      nv = 1;
    } else if (st[h] == GF_ST_PROBED) {
      nv = gf_sites(T, val[h], wi[h], v);
    }
    nvotes = gf_append_votes(S, nvotes, nv, v);
  }
  return nvotes;
}

// segment_mask (indexer.rs:616-679) over the class mask in S.u.p2.mask[0..L): longest run per
// target (3 = TOP with gp1, then 2 = SECOND with gp2), first start wins ties; writes the read's
// result.  S.u.p2.wcls is scratch.
// The reference tries every start s <= L-2 with mask[s] == target and scans forward,
// jumping gaps of up to 10 lower-class positions and stopping at a higher class.  Two
// facts make that parallel: (1) from any target position the scan continues the same
// way wherever it started, so a start that an earlier start's scan reaches can only give
// a shorter run with the same end — only "heads" (targets no earlier target reaches)
// matter; (2) the targets between two consecutive heads are exactly the first head's
// run, so its end is the last target before the next head.  Heads and targets are two
// bit masks (ballots, 64 positions per step); the few heads are then walked once.
template <int LCAP>
__device__ __forceinline__ void gf_segment_mask_wave(GfMapSmem<LCAP>& S, int L, int lane, int64_t gp1, int64_t gp2,
                                                     int64_t r, uint8_t* __restrict__ counts,
                                                     gf_seqmatch* __restrict__ matches, bool zc = false) {
  constexpr int NCH = (LCAP + 63) / 64;
  uint32_t* segT = (uint32_t*)S.u.p2.wcls;        // wcls is dead: reuse it for the masks (32-bit halves)
  uint32_t* segH = segT + 2 * NCH;
  static_assert(4 * NCH * 4 <= LCAP, "masks fit in wcls");
  auto ld64 = [](const uint32_t* a, int c) { return (uint64_t)a[2 * c] | ((uint64_t)a[2 * c + 1] << 32); };
  const int nch = (L + 63) >> 6;
  int nout = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int target = t == 0 ? 3 : 2;
    gf_wave_lds_sync();
    for (int c = 0; c < nch; ++c) {
      const int p = 64 * c + lane;
      const int m = p < L ? (int)S.u.p2.mask[p] : 255;
      const bool is_t = m == target;
      bool head = is_t;  // head of a chain: a target no earlier target's scan reaches
      if (head) {  // reached by an earlier target within the allowed gap, nothing higher in between?
        for (int k = 1; k <= GF_ALLOWED_GAP && p - k >= 0; ++k) {
          const int q = (int)S.u.p2.mask[p - k];
          if (q > target) break;
          if (q == target) {
            head = false;
            break;
          }
        }
      }
      const uint64_t bt = __ballot(is_t), bh = __ballot(head);
      if (lane == 0) {
        segT[2 * c] = (uint32_t)bt; segT[2 * c + 1] = (uint32_t)(bt >> 32);
        segH[2 * c] = (uint32_t)bh; segH[2 * c + 1] = (uint32_t)(bh >> 32);
      }
    }
    gf_wave_lds_sync();
    int best_len = -1, best_s = 0;
    for (int c = 0; c < nch; ++c) {
      uint64_t h = ld64(segH, c);
      while (h) {
        const int b = __builtin_ctzll(h);
        h &= h - 1;
        const int s = 64 * c + b;
        // the next head after s (L if there is none)
        int nh = L;
        if (h) {
          nh = 64 * c + __builtin_ctzll(h);
        } else {
          for (int c2 = c + 1; c2 < nch; ++c2) {
            const uint64_t h2 = ld64(segH, c2);
            if (h2) {
              nh = 64 * c2 + __builtin_ctzll(h2);
              break;
            }
          }
        }
        // the last target before it: s itself is one
        int e = s;
        for (int cw = (nh - 1) >> 6; cw >= c; --cw) {
          uint64_t tt = ld64(segT, cw);
          const int hi = nh - 64 * cw;  // keep bits < hi
          if (hi < 64) tt &= (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
          if (tt) {
            e = 64 * cw + 63 - __builtin_clzll(tt);
            break;
          }
        }
        if (s <= L - 2 && e - s > best_len) {  // (the last position is no start, :631)
          best_len = e - s;
          best_s = s;
        }
      }
    }
    if (best_len > GF_THRESHOLD_LEN) {
      if (lane == 0) {
        const int64_t gp = t == 0 ? gp1 : gp2;
        gf_seqmatch out;
        out.seq_start = best_s;
        out.seq_end = best_s + best_len;
        out.position = (int32_t)(uint32_t)(gp & 0xFFFFFFFFll);  // i64_to_gp, :709-714
        out.contig = (int16_t)(gp >> 32);
        out.pad = 0;
        matches[2 * r + nout] = out;
      }
      nout += 1;
    }
  }
  if (!zc) {
    if (lane == 0) counts[r] = (uint8_t)nout;
  } else if (nout > 0) {
    // a zero-copy host call (GfTable::done_*): counts start out zero in the host's pinned block and only a read WITH
    // segments writes anything; its wave then makes those few stores visible system-wide before it goes on, so that
    // the word the grid's last block stores cannot overtake them.  (A fence per wave, hits or not, made concurrent
    // callers' kernels take turns; waiting for the stores' acknowledgements alone let the word overtake them.)
    if (lane == 0) counts[r] = (uint8_t)nout;
    __threadfence_system();
  }
}

// ---- 3..5: peel, gate, second pass, segment_mask, output ----
template <int LCAP>
__device__ __forceinline__ void gf_finish_read(const GfTable& T, GfMapSmem<LCAP>& S, int L, uint32_t sh,
                                               int nvotes, int lane, int64_t r, uint8_t* __restrict__ counts,
                                               gf_seqmatch* __restrict__ matches) {
  // count1 >= 20 and count2 >= 10 on two different diagonals need >= 30 votes
  if (nvotes < (GF_MAJOR_KEYS + GF_MINOR_KEYS) / 2) {
    if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
    return;
  }
  gf_wave_lds_sync();

  // peel: count equal diagonals, keep the best two with >= 10 votes
  int64_t gp1 = 0, gp2 = 0;
  int cnt1 = 0, cnt2 = 0;
  {
    const int nchunks = (nvotes + 63) >> 6;
    int remaining = nvotes;
    for (int k = 0; k < nchunks && remaining >= GF_MINOR_KEYS / 2; ++k) {
      const int idx = (k << 6) + lane;
      uint32_t mine = (idx < nvotes) ? S.u.votes[idx] : GF_NONE_LIN;
      uint64_t alive = __ballot(mine != GF_NONE_LIN);
      while (alive != 0 && remaining >= GF_MINOR_KEYS / 2) {
        const int leader = __builtin_ctzll(alive);
        const uint32_t K = (uint32_t)__builtin_amdgcn_readlane((int)mine, leader);
        const bool eq0 = mine == K;
        int c = __popcll(__ballot(eq0));
        if (eq0) mine = GF_NONE_LIN;
        for (int kk = k + 1; kk < nchunks; ++kk) {  // later chunks: retired in place
          const int j = (kk << 6) + lane;
          const bool eq = (j < nvotes) && (S.u.votes[j] == K);
          c += __popcll(__ballot(eq));
          if (eq) S.u.votes[j] = GF_NONE_LIN;
        }
        remaining -= c;
        alive = __ballot(mine != GF_NONE_LIN);
        if (c >= GF_MINOR_KEYS / 2) {
          const int ctg = gf_contig_of_wave(T, K, lane);
          const int32_t d = (int32_t)(K - T.lin_base[ctg]);
          const int64_t key64 = (int64_t)(((uint64_t)(uint32_t)ctg << 32) | (uint64_t)(uint32_t)d);
          if (key64 != 0) {  // indexer.rs:337,342: key 0 never ranks
            if (c > cnt1 || (c == cnt1 && key64 < gp1)) {
              gp2 = gp1; cnt2 = cnt1; gp1 = key64; cnt1 = c;
            } else if (c > cnt2 || (c == cnt2 && key64 < gp2)) {
              gp2 = key64; cnt2 = c;
            }
          }
        }
      }
    }
  }
  if (cnt1 * 2 < GF_MAJOR_KEYS || cnt2 * 2 < GF_MINOR_KEYS) {
    if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
    return;
  }

  // second pass: classify every window (stride 1)
  uint32_t t1a = gf_lin_of_key(T, gp1 - 1), t1b = gf_lin_of_key(T, gp1), t1c = gf_lin_of_key(T, gp1 + 1);
  uint32_t t2a = gf_lin_of_key(T, gp2 - 1), t2b = gf_lin_of_key(T, gp2), t2c = gf_lin_of_key(T, gp2 + 1);
  const uint32_t lin0 = gf_lin_of_key(T, 0);
  gf_wave_lds_sync();  // votes are dead; wcls/mask reuse their LDS
  const int nwin2 = L - GF_KMER + 1;
  for (int w0 = 0; w0 < nwin2; w0 += 64) {
    const int w = w0 + lane;
    const bool act = w < nwin2;
    const uint32_t i = act ? (uint32_t)w : 0u;
    const uint32_t g = sh + i;
    const uint32_t key = gf_window(S.codes[g >> 4], S.codes[(g >> 4) + 1], g);
    const uint32_t bad = gf_flags16(S.inv[g >> 5], S.inv[(g >> 5) + 1], g);
    // A window whose 16 bases equal the bases of the site on gp1 (or gp2) at this offset, that site
    // being the only site of its key (gf_table.h: gdu), is known to return exactly that site from the
    // table: its class is 3 (or 2) and the probe is skipped.  On a read that passed the gate nearly
    // every window lies on one of the two diagonals; the others are probed as before.
    uint32_t cls = 0;
    bool known = !(act && !bad);
    if (!known) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const uint32_t tb = d == 0 ? t1b : t2b;
        const uint32_t a = tb + i;
        if (!known && tb != GF_NONE_LIN && (a >> 4) + 2 < T.gd_words) {
          const uint32_t gk = gf_window(T.gdu[2 * (a >> 4)], T.gdu[2 * (a >> 4) + 2], a);
          const uint32_t ubit = (T.gdu[2 * (a >> 4) + 1] >> (2u * (a & 15u))) & 1u;
          if (ubit && gk == key) {
            // the class of that one site's diagonal, by the reference's order of tests (gp2 within one
            // of gp1 — e.g. (c,-1) and (c+1,0), whose i64 keys are neighbours — is class 3, not 2)
            cls = (tb == t1a || tb == t1b || tb == t1c) ? 3u : ((tb == t2a || tb == t2b || tb == t2c) ? 2u : (tb == lin0 ? 1u : 0u));
            known = true;
          }
        }
      }
    }
    if (!known) {
      const uint32_t val = gf_lookup(T, key);
      uint32_t v[5];
      const int nv = gf_sites(T, val, i, v);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (k < nv) {
          const uint32_t x = v[k];
          uint32_t f = 0;
          if (x == t1a || x == t1b || x == t1c) f = 3;       // |gplong - gp1| <= 1
          else if (x == t2a || x == t2b || x == t2c) f = 2;  // |gplong - gp2| <= 1
          else if (x == lin0) f = 1;                         // gplong == 0
          cls = f > cls ? f : cls;
        }
      }
    }
    if (act) S.u.p2.wcls[w] = (uint8_t)cls;
  }
  gf_wave_lds_sync();

  // mask[j] = max class of the windows covering base j (make_mask, :716-732)
  int mismatches = 0;
  for (int j0 = 0; j0 < L; j0 += 64) {
    const int j = j0 + lane;
    uint32_t m = 0;
    if (j < L) {
      const int lo = j - (GF_KMER - 1) > 0 ? j - (GF_KMER - 1) : 0;
      const int hi = j < L - GF_KMER ? j : L - GF_KMER;
      for (int w = lo; w <= hi; ++w) {
        uint32_t c = S.u.p2.wcls[w];
        m = c > m ? c : m;
      }
      S.u.p2.mask[j] = (uint8_t)m;
    }
    mismatches += __popcll(__ballot(j < L && m <= 1));
  }
  if (mismatches > GF_MISMATCH_THRESHOLD) {
    if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
    return;
  }
  gf_wave_lds_sync();

  gf_segment_mask_wave<LCAP>(S, L, lane, gp1, gp2, r, counts, matches, T.done_flag != nullptr);
}

// End of a kernel of a zero-copy host call (GfTable::done_*): the block's results are made visible system-wide, the
// block counts itself out, the last block of the grid resets the counter and tells the host.  Every thread of the
// block reaches this (no early exits before it).
__device__ __forceinline__ void gf_block_done(const GfTable& T) {
  if (T.done_ctr == nullptr) return;  // (uniform)
  // (the waves that wrote results have fenced them at system scope where they wrote them, gf_segment_mask_wave)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = atomicAdd(T.done_ctr, 1u);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(T.done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the lane's next call finds it zero
      __threadfence_system();
      __hip_atomic_store(T.done_flag, T.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// PRODUCER: 0 = probe all windows, 1 = seed + verify (LCAP <= 256 only)
template <int LCAP, int WAVES, int PRODUCER>
__global__ __launch_bounds__(WAVES * 64) void gf_k_map_reads(GfTable T, const uint8_t* __restrict__ bases,
                                                             const int64_t* __restrict__ offsets,
                                                             int64_t n, int lmin, int mark_too_long,
                                                             uint8_t* __restrict__ counts,
                                                             gf_seqmatch* __restrict__ matches) {
  // handles the reads with lmin < length <= LCAP; shorter ones belong to another
  // launch of the same batch, longer ones too unless this is the top length class
  __shared__ GfMapSmem<LCAP> smem[WAVES];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  GfMapSmem<LCAP>& S = smem[wib];
  const int64_t stride = (int64_t)gridDim.x * WAVES;

  for (int64_t r = (int64_t)blockIdx.x * WAVES + wib; r < n; r += stride) {
    const int64_t off0 = offsets[r];
    const int64_t len64 = offsets[r + 1] - off0;
    if (T.skip != nullptr && T.skip[r] > 0) {  // not a candidate of this pass (gf_table.h: skip)
      if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
      continue;
    }
    if (len64 > LCAP) {
      if (mark_too_long && lane == 0) counts[r] = GF_COUNT_TOO_LONG;
      continue;
    }
    if (len64 <= lmin) continue;
    const int L = (int)len64;
    if (L < GF_KMER) {  // no window (also covers malformed negative lengths)
      if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
      continue;
    }
    gf_wave_lds_sync();  // previous read's LDS traffic is finished
    const uint32_t sh = gf_stage_read<LCAP>(S, bases + off0, L, lane);
    gf_wave_lds_sync();
    int nvotes;
    if constexpr (PRODUCER == 1) nvotes = gf_first_pass_seed_verify<LCAP>(T, S, L, sh, lane);
    else nvotes = gf_first_pass_probe_all<LCAP>(T, S, L, sh, lane);
    gf_finish_read<LCAP>(T, S, L, sh, nvotes, lane, r, counts, matches);
  }
  gf_block_done(T);
}

// The <= 256-base class with a two-deep software pipeline over the reads of a wave:
// while read k is processed, the bases of read k+1 and the offsets of read k+2 are
// already in flight, so the per-read dependent chain shrinks from
// offsets -> bases -> seeds -> verify/probe to seeds -> verify/probe.
template <int WAVES, int PRODUCER>
__global__ __launch_bounds__(WAVES * 64, 8) void gf_k_map_reads_short(GfTable T, const uint8_t* __restrict__ bases,
                                                                      const int64_t* __restrict__ offsets,
                                                                      int64_t n, int mark_too_long,
                                                                      uint8_t* __restrict__ counts,
                                                                      gf_seqmatch* __restrict__ matches) {
  constexpr int LCAP = 256;
  __shared__ GfMapSmem<LCAP> smem[WAVES];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  GfMapSmem<LCAP>& S = smem[wib];
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t r0 = (int64_t)blockIdx.x * WAVES + wib;
  if (r0 >= n) {  // (a wave without a read; it still counts its block out)
    gf_block_done(T);
    return;
  }

  // pipeline registers: [cur] bases loaded, [nxt] offsets loaded
  int64_t cur_off = offsets[r0], cur_end = offsets[r0 + 1];
  int64_t nxt_off = 0, nxt_end = 0;
  if (r0 + stride < n) {
    nxt_off = offsets[r0 + stride];
    nxt_end = offsets[r0 + stride + 1];
  }
  auto load_bases = [&](int64_t off, int64_t end, uint32_t& x0, uint32_t& x1) {
    const int64_t len = end - off;
    x0 = 0;
    x1 = 0;
    if (len >= GF_KMER && len <= LCAP) {
      const uintptr_t addr = (uintptr_t)(bases + off);
      const uint32_t sh = (uint32_t)(addr & 3u);
      const uint32_t* pw = (const uint32_t*)(addr - sh);
      const int ndw = (int)((sh + (uint32_t)len + 3u) >> 2);
      if (lane < ndw) x0 = pw[lane];
      if (lane + 64 < ndw) x1 = pw[lane + 64];
    }
  };
  uint32_t cur_x0, cur_x1;
  load_bases(cur_off, cur_end, cur_x0, cur_x1);

  for (int64_t r = r0; r < n; r += stride) {
    // issue the next read's bases and the offsets of the one after it
    uint32_t nxt_x0 = 0, nxt_x1 = 0;
    int64_t nn_off = 0, nn_end = 0;
    const bool have_next = r + stride < n;
    if (have_next) load_bases(nxt_off, nxt_end, nxt_x0, nxt_x1);
    if (r + 2 * stride < n) {
      nn_off = offsets[r + 2 * stride];
      nn_end = offsets[r + 2 * stride + 1];
    }

    const int64_t len64 = cur_end - cur_off;
    if (T.skip != nullptr && T.skip[r] > 0) {  // not a candidate of this pass (gf_table.h: skip)
      if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
    } else if (len64 > LCAP) {
      if (mark_too_long && lane == 0) counts[r] = GF_COUNT_TOO_LONG;
    } else if (len64 < GF_KMER) {
      if (lane == 0 && T.done_flag == nullptr) counts[r] = 0;  // (a zero-copy call's counts start out zero)
    } else {
      const int L = (int)len64;
      const uint32_t sh = (uint32_t)((uintptr_t)(bases + cur_off) & 3u);
      const int ndw = (int)((sh + (uint32_t)L + 3u) >> 2);
      gf_wave_lds_sync();  // previous read's LDS traffic is finished
      gf_stage_regs<LCAP>(S, cur_x0, cur_x1, ndw, lane);
      gf_wave_lds_sync();
      int nvotes;
#if defined(GF_ABLATE) && GF_ABLATE <= 1
      nvotes = -1;  // timing-only build: stop after staging the read in LDS
      if (S.codes[lane & 7] == 0x12345678u) nvotes = 1000;
#else
      if constexpr (PRODUCER == 1) nvotes = gf_first_pass_seed_verify<LCAP>(T, S, L, sh, lane);
      else nvotes = gf_first_pass_probe_all<LCAP>(T, S, L, sh, lane);
#endif
      gf_finish_read<LCAP>(T, S, L, sh, nvotes, lane, r, counts, matches);
    }
    cur_off = nxt_off; cur_end = nxt_end; cur_x0 = nxt_x0; cur_x1 = nxt_x1;
    nxt_off = nn_off; nxt_end = nn_end;
  }
  gf_block_done(T);
}

// Test/diagnostic: the device form of segment_mask alone, on class masks given by the caller (one
// wavefront per mask of lmin < length <= LCAP), so that it can be held against the reference's
// sequential scan on arbitrary masks and not only on the masks whole reads happen to produce.
template <int LCAP>
__global__ __launch_bounds__(64) void gf_k_segment_mask_test(const uint8_t* __restrict__ masks,
                                                             const int64_t* __restrict__ offsets, int64_t n, int lmin,
                                                             const int64_t* __restrict__ gp1,
                                                             const int64_t* __restrict__ gp2,
                                                             uint8_t* __restrict__ counts,
                                                             gf_seqmatch* __restrict__ matches) {
  __shared__ GfMapSmem<LCAP> S;
  const int lane = threadIdx.x & 63;
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    const int64_t off0 = offsets[r];
    const int64_t len64 = offsets[r + 1] - off0;
    if (len64 <= lmin || len64 > LCAP) continue;
    const int L = (int)len64;
    gf_wave_lds_sync();
    for (int j = lane; j < L; j += 64) S.u.p2.mask[j] = masks[off0 + j];
    gf_wave_lds_sync();
    gf_segment_mask_wave<LCAP>(S, L, lane, gp1[r], gp2[r], r, counts, matches);
  }
}

// Test/diagnostic: k-mer -> stored sites, decoded to (contig, position).
__global__ void gf_k_lookup(GfTable T, const uint32_t* __restrict__ ref_kmers, int64_t n,
                            int32_t* __restrict__ out_count, int16_t* __restrict__ out_contig,
                            int32_t* __restrict__ out_position) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t val = gf_lookup(T, gf_key_from_ref_kmer(ref_kmers[q]));
    uint32_t type = val >> GF_TYPE_SHIFT;
    if (type == GF_TYPE_HIGH) {
      out_count[q] = -2;
      continue;
    }
    uint32_t v[5];
    int nv = gf_sites(T, val, 0u, v);
    out_count[q] = nv;
    for (int k = 0; k < nv; ++k) {
      int c = gf_contig_of(T, v[k]);
      out_contig[5 * q + k] = (int16_t)c;
      out_position[5 * q + k] = (int32_t)(v[k] - T.lin_base[c]);
    }
  }
}
