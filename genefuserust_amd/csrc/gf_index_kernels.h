// K1 — index build on the device.
//
// Reproduces Indexer::make_index + index_contig + fill_bloom_filter
// (src/core/indexer.rs:122-250) as an order-independent classification
// (SURVEY.md Appendix A.1): every valid 16-base window of every gene yields a
// forward site (position f, windows 0..len-17) and a reverse-complement site
// (position -(f+15), windows 1..len-16; the reverse window i of indexer.rs:168
// is the forward window f = len-16-i, so position = i+1-len = -(f+15)).
// Keys seen once keep their site, 2..5 times keep all sites, >= 6 times become
// HIGH.  No sort.  The presence filter by hash partitions (gf_k_filter_scatter / gf_k_filter_build: no global
// atomics), one pass over the gene bases (gf_k_index_insert: a key's first site is written with the
// claim of its slot, later sites go to a side list), a sweep over the table (counters -> unique /
// dupes(start in dupes[]) / HIGH, statistics), the side list into the duplicate lists (each sorted by the thread
// that brings its last site); the same pass over the side list flags every site of a HIGH key (the odd bit beside its
// "unique" flag in gdu) and leaves the smallest of them in the key's slot (r04: what the mapping kernels prove "cannot
// vote" from).  The first form is kept behind GF_BUILD_TWO_PASS (experiments; it keeps neither):
//   COUNT   insert keys with 64-bit CAS, count occurrences
//   classify (count -> unique / dupes(start in dupes[]) / HIGH)
//   FILL    write site codes
// Slot placement inside a bucket depends on the race order of different keys, so
// the table's byte image is not reproducible run to run; every lookup result is.
#pragma once

#include <hip/hip_runtime.h>

#include "gf_table.h"

#define GF_TILE_BASES 4096
#define GF_INDEX_THREADS 256

struct GfGenes {
  const uint8_t* cat;       // upper-cased gene bytes, concatenated, zero padded (>= 32 B)
  const uint32_t* gene_off; // [n_genes + 1] start of each gene in cat
  const uint32_t* lin_base; // [n_genes]
  uint32_t total;           // bytes in cat
  int32_t n_genes;
};

enum { GF_MODE_COUNT = 0, GF_MODE_FILL = 1 };

// indexer.rs:159: the gene slices are upper-cased before they are indexed (ASCII letters; the FASTA reader keeps
// letters, '-' and '*' only).  In place on the device copy, 16 bytes per thread and step.
__global__ __launch_bounds__(256) void gf_k_upper_inplace(uint8_t* __restrict__ p, unsigned long long n_vec) {
  uint4* v = (uint4*)p;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (unsigned long long)gridDim.x * 256) {
    uint4 q = v[i];
    uint32_t w[4] = {q.x, q.y, q.z, q.w};
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k], y = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t ch = (x >> (8 * b)) & 0xFFu;
        if (ch - 'a' < 26u) ch -= 32u;
        y |= ch << (8 * b);
      }
      any |= y != x;
      w[k] = y;
    }
    if (any) v[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// The same out of place, `src` being the host's pinned staging block: the kernel pulls a chunk of the gene bytes
// over the bus itself as soon as the host has gathered it (r03: a DMA per megabyte cost 40 us of set-up each, a
// launch of this costs 5).
__global__ __launch_bounds__(256) void gf_k_upper_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       unsigned long long n_vec) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t* v = (const u32x4_t*)src;
  uint4* o = (uint4*)dst;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (unsigned long long)gridDim.x * 256) {
    const u32x4_t q = __builtin_nontemporal_load(v + i);
    uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k], y = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t ch = (x >> (8 * b)) & 0xFFu;
        if (ch - 'a' < 26u) ch -= 32u;
        y |= ch << (8 * b);
      }
      w[k] = y;
    }
    o[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__device__ __forceinline__ uint64_t gf_atomic_load64(uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// COUNT pass: claim a slot for `key` (or find it) and bump its occurrence count.
// A bucket's slots are claimed in order (a slot is taken only by a thread that saw every slot before it
// occupied), so the occupied slots are a prefix.  The eight slots are read at once — one round trip to the
// bucket's 64-byte line in place of one per occupied slot — and the slot-by-slot search starts at the first
// slot that was still empty in that snapshot; keys never change or leave, so a key seen there is there.
__device__ __forceinline__ void gf_insert_count(uint64_t* slots, uint32_t nbuckets, uint32_t key) {
  uint32_t b = gf_bucket_of(key, nbuckets);
  for (uint32_t guard = 0; guard <= nbuckets; ++guard) {
    uint64_t* bucket = slots + (size_t)b * GF_SLOTS_PER_BUCKET;
    // (plain loads: a stale line shows an earlier state of the bucket — fewer slots taken, never a wrong key —
    //  and the CAS below finds out what a slot holds by now)
    const uint4* q = (const uint4*)bucket;
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const uint32_t lo[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
    const uint32_t hi[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
    int hit = -1, first_empty = GF_SLOTS_PER_BUCKET;
#pragma unroll
    for (int j = GF_SLOTS_PER_BUCKET - 1; j >= 0; --j) {
      if ((lo[j] & GF_VAL_LOW) == 0) first_empty = j;
      else if (hi[j] == key) hit = j;
    }
    if (hit >= 0) {
      atomicAdd((unsigned int*)(bucket + hit), 1u);  // low word = count
      return;
    }
    for (int j = first_empty; j < GF_SLOTS_PER_BUCKET; ++j) {
      uint64_t cur = j == first_empty ? 0ull : gf_atomic_load64(bucket + j);
      if ((cur & GF_VAL_LOW) == 0) {
        uint64_t want = ((uint64_t)key << 32) | 1ull;
        uint64_t prev = atomicCAS((unsigned long long*)(bucket + j), 0ull, (unsigned long long)want);
        if (prev == 0) return;
        cur = prev;
      }
      if ((uint32_t)(cur >> 32) == key) {
        atomicAdd((unsigned int*)(bucket + j), 1u);
        return;
      }
    }
    atomicOr((unsigned int*)bucket, GF_VAL_OVF);  // slot 0, low word
    b = (b + 1 == nbuckets) ? 0 : b + 1;
  }
}

// Slot holding `key` (must have been inserted; the table no longer changes its keys).
__device__ __forceinline__ uint64_t* gf_find_slot(uint64_t* slots, uint32_t nbuckets, uint32_t key) {
  uint32_t b = gf_bucket_of(key, nbuckets);
  for (uint32_t guard = 0; guard <= nbuckets; ++guard) {
    uint64_t* bucket = slots + (size_t)b * GF_SLOTS_PER_BUCKET;
    const uint4* q = (const uint4*)bucket;  // the whole 64-byte line, four loads in flight
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const uint32_t lo[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
    const uint32_t hi[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
    int hit = -1;
#pragma unroll
    for (int j = GF_SLOTS_PER_BUCKET - 1; j >= 0; --j)
      if ((lo[j] & GF_VAL_LOW) != 0 && hi[j] == key) hit = j;
    if (hit >= 0) return bucket + hit;
    if (!(lo[0] & GF_VAL_OVF)) return nullptr;
    b = (b + 1 == nbuckets) ? 0 : b + 1;
  }
  return nullptr;
}

// FILL pass: store the site code of one occurrence; true when the key has exactly this one site (the
// caller then sets the site's "unique" flag in gdu, a wavefront's flags at a time).
__device__ __forceinline__ bool gf_fill_site(uint64_t* slots, uint32_t nbuckets, uint32_t* dupes, uint32_t key, uint32_t lin) {
  uint64_t* s = gf_find_slot(slots, nbuckets, key);
  if (!s) return false;
  uint32_t* valp = (uint32_t*)s;  // little-endian: low word = val
  uint32_t val = *valp;
  uint32_t type = (val & GF_VAL_LOW) >> GF_TYPE_SHIFT;
  if (type == GF_TYPE_UNIQUE) {
    // exactly one occurrence => exactly one writer
    *valp = (val & GF_VAL_OVF) | (GF_TYPE_UNIQUE << GF_TYPE_SHIFT) | (lin & GF_LIN_MASK);
    return true;
  } else if (type == GF_TYPE_DUPES) {
    uint32_t cnt = (val >> GF_DUPE_COUNT_SHIFT) & 7u;
    uint32_t start = val & GF_DUPE_START_MASK;
    for (uint32_t k = 0; k < cnt; ++k)
      if (atomicCAS(dupes + start + k, GF_DUPE_EMPTY, lin) == GF_DUPE_EMPTY) break;
  }
  return false;
}

// The "unique" flags of a wavefront's sites (flag of site code lin: gdu[2 * (lin >> 4) + 1], bit 2 * (lin & 15)).
// The lanes of a wavefront hold consecutive site codes, 16 to a flag word: the lanes of a word OR their bits
// together with shuffles and the first of them issues the one atomic (an atomic per site was 29 M of them on a
// cancer-sized gene set, sixteen neighbours to the same address).  Called by all 64 lanes.
__device__ __forceinline__ void gf_wave_set_unique(uint32_t* gdu, bool has, uint32_t lin) {
  const int lane = threadIdx.x & 63;
  const uint32_t wi = has ? (lin >> 4) : 0xFFFFFFFFu;
  uint32_t v = has ? 1u << (2u * (lin & 15u)) : 0u;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const uint32_t y = __shfl_down(v, o), wy = __shfl_down(wi, o);
    if (lane + o < 64 && wy == wi) v |= y;
  }
  const uint32_t wp = __shfl_up(wi, 1);
  if (v && (lane == 0 || wp != wi)) atomicOr(gdu + 2 * (size_t)wi + 1, v);
}

// One canonical 14-mer into the presence filter: most bit pairs of a repeated 14-mer are set already, so
// look before the atomic.
__device__ __forceinline__ void gf_bloom_insert(uint32_t* bloom, uint32_t nwords, uint32_t s14) {
  const uint32_t h = GF_BLOOM_HASH((s14));
  uint32_t* w = bloom + GF_BLOOM_WORD(h, nwords);
  const uint32_t b = GF_BLOOM_BITS(h);
  if ((__builtin_nontemporal_load(w) & b) != b) atomicOr(w, b);
}

// One block = one tile of GF_TILE_BASES window starts.  Phase 1 packs the tile's
// ASCII bases (+ halo) into a 2-bit stream and an invalid-bit stream in LDS with
// coalesced 16-byte loads; phase 2 cuts the windows out of LDS.
// The FILL pass also sets the sites' "unique" flags and fills the presence filter (bloom != nullptr): every key's
// first and last 14 bases in canonical form (gf_table.h) — a window enters its first 14 bases, and its last 14
// only where window + 2, whose first 14 they are, has no key; both strands of a window share the canonical forms.
template <int MODE>
__global__ __launch_bounds__(GF_INDEX_THREADS) void gf_k_index_sites(GfGenes G, uint64_t* slots,
                                                                     uint32_t nbuckets,
                                                                     uint32_t* dupes, uint32_t* gdu,
                                                                     uint32_t* bloom, uint32_t bloom_words) {
  __shared__ uint32_t s_codes[GF_TILE_BASES / 16 + 2];
  __shared__ uint32_t s_inv[GF_TILE_BASES / 32 + 2];
  const uint32_t t0 = blockIdx.x * GF_TILE_BASES;
  const int tid = threadIdx.x;

  for (int ch = tid; ch < GF_TILE_BASES / 16 + 1; ch += GF_INDEX_THREADS) {
    // cat is padded so that this 16-byte load stays inside the allocation
    uint4 q = *(const uint4*)(G.cat + (size_t)t0 + 16u * ch);
    uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
    gf_convert4(q.x, c0, i0);
    gf_convert4(q.y, c1, i1);
    gf_convert4(q.z, c2, i2);
    gf_convert4(q.w, c3, i3);
    s_codes[ch] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    ((uint16_t*)s_inv)[ch] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
  }
  if (tid == 0) {
    s_codes[GF_TILE_BASES / 16 + 1] = 0;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 1] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 2] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 3] = 0xFFFF;
  }
  __syncthreads();

  for (int l0 = 0; l0 < GF_TILE_BASES; l0 += GF_INDEX_THREADS) {  // (every lane of a wave stays in the loop)
    const int l = l0 + tid;
    const uint32_t g = t0 + (uint32_t)l;
    bool fwd = false, rev = false, next_has = false;
    uint32_t key = 0, lin_f = 0, lin_r = 0;
    if (g < G.total) {
      // the invalid-base flags of bases l .. l+31 (the halo chunk beyond the tile is flagged invalid from
      // base TILE + 16 on, the one after it entirely)
      const uint32_t sh = (uint32_t)l & 31u;
      const uint32_t lo_w = s_inv[l >> 5], hi_w = s_inv[(l >> 5) + 1];
      const uint32_t inv = sh ? ((lo_w >> sh) | (hi_w << (32u - sh))) : lo_w;
      if ((inv & 0xFFFFu) == 0) {
        // gene of g: last c with gene_off[c] <= g
        int lo = 0, hi = G.n_genes;  // invariant gene_off[lo] <= g < gene_off[hi]
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (G.gene_off[mid] <= g) lo = mid; else hi = mid;
        }
        const uint32_t f = g - G.gene_off[lo];
        const uint32_t len = G.gene_off[lo + 1] - G.gene_off[lo];
        if (f + GF_KMER <= len) {  // the window lies inside the gene
          key = gf_window(s_codes[l >> 4], s_codes[(l >> 4) + 1], (uint32_t)l);
          fwd = f + GF_KMER < len;  // forward windows 0 .. len-17 (indexer.rs:188)
          rev = f >= 1;             // reverse windows i = len-16-f in 0 .. len-17
          lin_f = G.lin_base[lo] + f;
          lin_r = G.lin_base[lo] - (f + 15u);
          // window + 2 has a key of its own (it then enters the 14 bases the two share).  In the tile's last
          // two windows the flags of its last bases are beyond the halo: "no" there costs one more look-up.
          next_has = ((inv >> 2) & 0xFFFFu) == 0 && f + 2 + GF_KMER <= len && l + 2 + GF_KMER <= GF_TILE_BASES + 16;
        }
      }
    }
    if (MODE == GF_MODE_COUNT) {
      if (fwd) gf_insert_count(slots, nbuckets, key);
      if (rev) gf_insert_count(slots, nbuckets, gf_revcomp_key(key));
    } else {
      const bool uf = fwd && gf_fill_site(slots, nbuckets, dupes, key, lin_f);
      const bool ur = rev && gf_fill_site(slots, nbuckets, dupes, gf_revcomp_key(key), lin_r);
      gf_wave_set_unique(gdu, uf, lin_f);
      gf_wave_set_unique(gdu, ur, lin_r);
      if (bloom && (fwd || rev)) {
        gf_bloom_insert(bloom, bloom_words, key & 0x0FFFFFFFu);
        if (!next_has) gf_bloom_insert(bloom, bloom_words, key >> 4);
      }
    }
  }
}

// ---- the one-pass build: claim and site in one touch of the bucket ----
// 97 % of the keys of a gene set have one site.  The COUNT / FILL pair touches every site's bucket twice — both
// random 64-byte lines far from every cache — to learn the count before a site may be written.  Here the thread
// that claims a slot writes its site with the claim (the final form of a key with one site); a thread that finds
// its key already there turns the slot into a counter (the first site moves to a side list with the thread's
// own) or bumps it and adds its site to the list.  A sweep then hands out the duplicate lists, and the side list
// — the 3 % — fills them.  "Unique" flags are set with the claim and cleared again for every site on the list.
struct GfSideEntry { uint32_t key, lin; };

// true: the slot was claimed for (key, lin), which is so far the key's only site; false: the key was there, the
// site (and, for the thread that found the key with one site, that site) went to e0 / e1, ne = how many
__device__ __forceinline__ bool gf_insert_site(uint64_t* slots, uint32_t nbuckets, uint32_t key, uint32_t lin,
                                               GfSideEntry& e0, GfSideEntry& e1, int& ne) {
  uint32_t b = gf_bucket_of(key, nbuckets);
  const uint32_t mine = (GF_TYPE_UNIQUE << GF_TYPE_SHIFT) | (lin & GF_LIN_MASK);
  ne = 0;
  for (uint32_t guard = 0; guard <= nbuckets; ++guard) {
    uint64_t* bucket = slots + (size_t)b * GF_SLOTS_PER_BUCKET;
    const uint4* q = (const uint4*)bucket;  // a snapshot (see gf_insert_count)
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const uint32_t lo[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
    const uint32_t hi[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
    int at = -1, first_empty = GF_SLOTS_PER_BUCKET;
#pragma unroll
    for (int j = GF_SLOTS_PER_BUCKET - 1; j >= 0; --j) {
      if ((lo[j] & GF_VAL_LOW) == 0) first_empty = j;
      else if (hi[j] == key) at = j;
    }
    if (at < 0) {
      for (int j = first_empty; j < GF_SLOTS_PER_BUCKET; ++j) {
        uint64_t cur = j == first_empty ? 0ull : gf_atomic_load64(bucket + j);
        if ((cur & GF_VAL_LOW) == 0) {
          const uint64_t want = ((uint64_t)key << 32) | mine;
          const uint64_t prev = atomicCAS((unsigned long long*)(bucket + j), 0ull, (unsigned long long)want);
          if (prev == 0) return true;
          cur = prev;
        }
        if ((uint32_t)(cur >> 32) == key) {
          at = j;
          break;
        }
      }
      if (at < 0) {
        atomicOr((unsigned int*)bucket, GF_VAL_OVF);  // slot 0, low word
        b = (b + 1 == nbuckets) ? 0 : b + 1;
        continue;
      }
    }
    // a later site of a key that is there
    unsigned int* lop = (unsigned int*)(bucket + at);
    uint32_t cur = __hip_atomic_load(lop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      if (((cur & GF_VAL_LOW) >> GF_TYPE_SHIFT) == GF_TYPE_UNIQUE) {
        const uint32_t want = (cur & GF_VAL_OVF) | 2u;  // counter form: two sites
        const uint32_t prev = atomicCAS(lop, cur, want);
        if (prev == cur) {
          e0.key = key; e0.lin = cur & GF_LIN_MASK;
          e1.key = key; e1.lin = lin;
          ne = 2;
          return false;
        }
        cur = prev;  // turned into a counter by somebody else, or the bucket's overflow flag was set meanwhile
      } else {
        atomicAdd(lop, 1u);
        e0.key = key; e0.lin = lin;
        ne = 1;
        return false;
      }
    }
  }
  return false;
}

#define GF_SIDE_LDS 2048  // side-list entries a block gathers before it takes room in the global list (one atomic)

__device__ __forceinline__ unsigned int gf_ix_wave_append_lds(bool want, unsigned int* s_counter) {
  const uint64_t m = __ballot(want);
  unsigned int base = 0;
  if (m) {
    const int leader = __builtin_ctzll(m);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(s_counter, (unsigned int)__popcll(m));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
  }
  return base + (unsigned int)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// the tile loop of gf_k_index_sites with gf_insert_site; side / side_n: the global list and its fill (entries
// beyond side_cap are counted, not written: the host checks the count)
__global__ __launch_bounds__(GF_INDEX_THREADS) void gf_k_index_insert(GfGenes G, uint64_t* slots, uint32_t nbuckets,
                                                                      uint32_t* gdu, uint32_t* bloom,
                                                                      uint32_t bloom_words, GfSideEntry* side,
                                                                      unsigned long long* side_n,
                                                                      unsigned long long side_cap) {
  __shared__ uint32_t s_codes[GF_TILE_BASES / 16 + 2];
  __shared__ uint32_t s_inv[GF_TILE_BASES / 32 + 2];
  __shared__ GfSideEntry s_side[GF_SIDE_LDS];
  __shared__ unsigned int s_fill;
  __shared__ unsigned long long s_base;
  const uint32_t t0 = blockIdx.x * GF_TILE_BASES;
  const int tid = threadIdx.x;
  for (int ch = tid; ch < GF_TILE_BASES / 16 + 1; ch += GF_INDEX_THREADS) {
    uint4 q = *(const uint4*)(G.cat + (size_t)t0 + 16u * ch);
    uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
    gf_convert4(q.x, c0, i0);
    gf_convert4(q.y, c1, i1);
    gf_convert4(q.z, c2, i2);
    gf_convert4(q.w, c3, i3);
    s_codes[ch] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    ((uint16_t*)s_inv)[ch] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
  }
  if (tid == 0) {
    s_codes[GF_TILE_BASES / 16 + 1] = 0;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 1] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 2] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 3] = 0xFFFF;
    s_fill = 0;
  }
  __syncthreads();
  auto flush = [&]() {  // (called by the whole block, between barriers)
    const unsigned int nf = s_fill;
    if (tid == 0 && nf) s_base = atomicAdd(side_n, (unsigned long long)nf);
    __syncthreads();
    for (unsigned int i = tid; i < nf; i += GF_INDEX_THREADS)
      if (s_base + i < side_cap) side[s_base + i] = s_side[i];
    __syncthreads();
    if (tid == 0) s_fill = 0;
    __syncthreads();
  };
  for (int l0 = 0; l0 < GF_TILE_BASES; l0 += GF_INDEX_THREADS) {  // (every lane of a wave stays in the loop)
    const int l = l0 + tid;
    const uint32_t g = t0 + (uint32_t)l;
    bool fwd = false, rev = false, next_has = false;
    uint32_t key = 0, lin_f = 0, lin_r = 0;
    if (g < G.total) {
      const uint32_t sh = (uint32_t)l & 31u;
      const uint32_t lo_w = s_inv[l >> 5], hi_w = s_inv[(l >> 5) + 1];
      const uint32_t inv = sh ? ((lo_w >> sh) | (hi_w << (32u - sh))) : lo_w;
      if ((inv & 0xFFFFu) == 0) {
        int lo = 0, hi = G.n_genes;  // invariant gene_off[lo] <= g < gene_off[hi]
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (G.gene_off[mid] <= g) lo = mid; else hi = mid;
        }
        const uint32_t f = g - G.gene_off[lo];
        const uint32_t len = G.gene_off[lo + 1] - G.gene_off[lo];
        if (f + GF_KMER <= len) {
          key = gf_window(s_codes[l >> 4], s_codes[(l >> 4) + 1], (uint32_t)l);
          fwd = f + GF_KMER < len;
          rev = f >= 1;
          lin_f = G.lin_base[lo] + f;
          lin_r = G.lin_base[lo] - (f + 15u);
          next_has = ((inv >> 2) & 0xFFFFu) == 0 && f + 2 + GF_KMER <= len && l + 2 + GF_KMER <= GF_TILE_BASES + 16;
        }
      }
    }
    GfSideEntry ef0{0, 0}, ef1{0, 0}, er0{0, 0}, er1{0, 0};
    int nf = 0, nr = 0;
    const bool cf = fwd && gf_insert_site(slots, nbuckets, key, lin_f, ef0, ef1, nf);
    const bool cr = rev && gf_insert_site(slots, nbuckets, gf_revcomp_key(key), lin_r, er0, er1, nr);
    gf_wave_set_unique(gdu, cf, lin_f);
    gf_wave_set_unique(gdu, cr, lin_r);
    if (bloom && (fwd || rev)) {
      gf_bloom_insert(bloom, bloom_words, key & 0x0FFFFFFFu);
      if (!next_has) gf_bloom_insert(bloom, bloom_words, key >> 4);
    }
    // the side entries of this round into the block's buffer (at most four per thread: it has room for them)
    {
      unsigned int at;
      at = gf_ix_wave_append_lds(nf > 0, &s_fill); if (nf > 0) s_side[at] = ef0;
      at = gf_ix_wave_append_lds(nf > 1, &s_fill); if (nf > 1) s_side[at] = ef1;
      at = gf_ix_wave_append_lds(nr > 0, &s_fill); if (nr > 0) s_side[at] = er0;
      at = gf_ix_wave_append_lds(nr > 1, &s_fill); if (nr > 1) s_side[at] = er1;
    }
    // (a barrier that ORs the threads' views: one that looks early may see less than the round's total, the
    //  last one to append sees it all, and every thread gets the same answer)
    if (__syncthreads_or(s_fill > GF_SIDE_LDS - 4 * GF_INDEX_THREADS)) flush();
  }
  __syncthreads();
  flush();
}

// The side list: every site of every key with more than one.  Keys with 2..5 sites get them into their list
// (assigned by the sweep); every listed site loses the "unique" flag its claim may have set.
__global__ void gf_k_index_side(const GfSideEntry* __restrict__ side, unsigned long long n, uint64_t* slots,
                                uint32_t nbuckets, uint32_t* dupes, uint32_t* gdu) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const GfSideEntry e = side[i];
    atomicAnd(gdu + 2 * (size_t)(e.lin >> 4) + 1, ~(1u << (2u * (e.lin & 15u))));
    uint64_t* s = gf_find_slot(slots, nbuckets, e.key);
    if (!s) continue;
    const uint32_t val = *(const uint32_t*)s;
    if (((val & GF_VAL_LOW) >> GF_TYPE_SHIFT) == GF_TYPE_HIGH) {
      // r04: a key with six sites or more cannot vote (indexer.rs:202-239), and a read that lies in a repeat shows
      // nothing else — the bucket pass probed fifty of its windows one after the other to learn it.  Every site of
      // such a key gets a flag (the odd bit beside its "unique" flag) and the key's slot the smallest of its sites:
      // one probe that comes back HIGH then leads to a copy of the repeat in the genes, and every window of the
      // read that equals a flagged site there is PROVEN unable to vote (gf_k_probe_buckets).
      atomicOr(gdu + 2 * (size_t)(e.lin >> 4) + 1, 2u << (2u * (e.lin & 15u)));
      atomicMin((unsigned int*)s, (val & ~GF_LIN_MASK) | (e.lin & GF_LIN_MASK));  // (the bits above the site field are final here)
      continue;
    }
    if (((val & GF_VAL_LOW) >> GF_TYPE_SHIFT) != GF_TYPE_DUPES) continue;
    const uint32_t cnt = (val >> GF_DUPE_COUNT_SHIFT) & 7u, start = val & GF_DUPE_START_MASK;
    for (uint32_t k = 0; k < cnt; ++k) {
      if (atomicCAS(dupes + start + k, GF_DUPE_EMPTY, e.lin) != GF_DUPE_EMPTY) continue;
      // A list fills from its front (a thread takes entry k only after it saw 0 .. k-1 taken), so whoever takes
      // the LAST entry knows the list complete and sorts it here, ascending (reproducible content) — the sweep
      // over the whole table that did this (gf_k_sort_dupes) was 0.1 ms of a cancer-sized build.
      if (k == cnt - 1 && cnt > 1) {
        uint32_t d[GF_DUP_THRESHOLD];
#pragma unroll
        for (uint32_t a = 0; a < GF_DUP_THRESHOLD; ++a)
          d[a] = a + 1 < cnt ? __hip_atomic_load(dupes + start + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : (a + 1 == cnt ? e.lin : 0xFFFFFFFFu);
#pragma unroll
        for (int a = 1; a < (int)GF_DUP_THRESHOLD; ++a)  // (insertion sort of five registers; the unused ones are maximal)
#pragma unroll
          for (int b = a; b > 0; --b)
            if (d[b - 1] > d[b]) { const uint32_t t = d[b]; d[b] = d[b - 1]; d[b - 1] = t; }
#pragma unroll
        for (uint32_t a = 0; a < GF_DUP_THRESHOLD; ++a)
          if (a < cnt) __hip_atomic_store(dupes + start + a, d[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
  }
}

// ---- the presence filter without global atomics (r03) ----
// Device-scope atomics are carried out beyond the L2s on this chip, about 22 G of them a second whatever they
// touch (tools/exp_build_costs.hip: 45 ps per CAS into the table, 55 ps per look-then-atomicOr into a 7.7 MiB
// filter), and gf_k_index_insert is exactly their sum: 30 M CAS + 15 M filter ORs = 2.05 ms on a cancer-sized gene
// set.  The filter's share goes: the hashes are scattered into partitions of GF_FSLICE_WORDS filter words (the
// word index is monotone in the hash, so a partition is a hash range), a block per partition ORs its hashes into
// LDS and writes its 32 KB of filter once.  A partition that outgrows its room (a gene set of one repeated
// 14-mer) sends the excess through the atomic as before — which the building block ORs over, not overwrites.
#define GF_FSLICE_WORDS 8192u
#define GF_FPARTS_MAX 256u

__global__ __launch_bounds__(GF_INDEX_THREADS) void gf_k_filter_scatter(GfGenes G, uint32_t* bloom, uint32_t bloom_words,
                                                                        uint32_t nparts, uint32_t* part_buf,
                                                                        uint32_t part_cap, unsigned int* part_fill) {
  __shared__ uint32_t s_codes[GF_TILE_BASES / 16 + 2];
  __shared__ uint32_t s_inv[GF_TILE_BASES / 32 + 2];
  __shared__ uint32_t s_h[2 * GF_TILE_BASES];
  __shared__ unsigned int s_cnt[GF_FPARTS_MAX], s_base[GF_FPARTS_MAX];
  __shared__ unsigned int s_n;
  __shared__ uint32_t s_goff[64];
  const uint32_t t0 = blockIdx.x * GF_TILE_BASES;
  const int tid = threadIdx.x;
  for (int ch = tid; ch < GF_TILE_BASES / 16 + 1; ch += GF_INDEX_THREADS) {
    uint4 q = *(const uint4*)(G.cat + (size_t)t0 + 16u * ch);
    uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
    gf_convert4(q.x, c0, i0);
    gf_convert4(q.y, c1, i1);
    gf_convert4(q.z, c2, i2);
    gf_convert4(q.w, c3, i3);
    s_codes[ch] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    ((uint16_t*)s_inv)[ch] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
  }
  if (tid == 0) {
    s_codes[GF_TILE_BASES / 16 + 1] = 0;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 1] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 2] = 0xFFFF;
    ((uint16_t*)s_inv)[GF_TILE_BASES / 16 + 3] = 0xFFFF;
    s_n = 0;
  }
  for (int p = tid; p < (int)GF_FPARTS_MAX; p += GF_INDEX_THREADS) s_cnt[p] = 0;
  // the genes this tile touches, found once (a tile is a fifth of an average gene: one or two of them; a search of
  // the offsets per base, ten dependent loads, was most of this kernel's time)
  if (tid < 64) {
    int lo = 0, hi = G.n_genes;  // invariant gene_off[lo] <= t0 < gene_off[hi]  (t0 < total)
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (G.gene_off[mid] <= t0) lo = mid; else hi = mid;
    }
    const int c = lo + tid;
    s_goff[tid] = c <= G.n_genes ? G.gene_off[c] : 0xFFFFFFFFu;
  }
  __syncthreads();
  // (a tile of more than 62 genes — slices of a few dozen bases — goes by the search after all)
  const bool few = s_goff[63] == 0xFFFFFFFFu || s_goff[63] >= t0 + GF_TILE_BASES + GF_KMER;
  // the tile's hashes, in any order: a window with a key enters its first 14 bases, and its last 14 where
  // window + 2 has no key of its own (gf_k_index_insert's rule, word for word)
  for (int l0 = 0; l0 < GF_TILE_BASES; l0 += GF_INDEX_THREADS) {  // (every lane of a wave stays in the loop)
    const int l = l0 + tid;
    const uint32_t g = t0 + (uint32_t)l;
    bool has = false, next_has = false;
    uint32_t key = 0;
    if (g < G.total) {
      const uint32_t sh = (uint32_t)l & 31u;
      const uint32_t lo_w = s_inv[l >> 5], hi_w = s_inv[(l >> 5) + 1];
      const uint32_t inv = sh ? ((lo_w >> sh) | (hi_w << (32u - sh))) : lo_w;
      if ((inv & 0xFFFFu) == 0) {
        uint32_t f, len;
        if (few) {
          int k = 0;
          while (s_goff[k + 1] <= g) ++k;  // (gene_off[n_genes] = total > g ends it)
          f = g - s_goff[k];
          len = s_goff[k + 1] - s_goff[k];
        } else {
          int lo = 0, hi = G.n_genes;  // invariant gene_off[lo] <= g < gene_off[hi]
          while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (G.gene_off[mid] <= g) lo = mid; else hi = mid;
          }
          f = g - G.gene_off[lo];
          len = G.gene_off[lo + 1] - G.gene_off[lo];
        }
        if (f + GF_KMER <= len) {
          key = gf_window(s_codes[l >> 4], s_codes[(l >> 4) + 1], (uint32_t)l);
          has = f + GF_KMER < len || f >= 1;  // a forward or a reverse site
          next_has = ((inv >> 2) & 0xFFFFu) == 0 && f + 2 + GF_KMER <= len && l + 2 + GF_KMER <= GF_TILE_BASES + 16;
        }
      }
    }
    const bool two = has && !next_has;
    unsigned int at = gf_ix_wave_append_lds(has, &s_n);
    if (has) s_h[at] = GF_BLOOM_HASH((key & 0x0FFFFFFFu));
    at = gf_ix_wave_append_lds(two, &s_n);
    if (two) s_h[at] = GF_BLOOM_HASH((key >> 4));
  }
  __syncthreads();
  const unsigned int n = s_n;
  for (unsigned int i = tid; i < n; i += GF_INDEX_THREADS)
    atomicAdd(&s_cnt[GF_BLOOM_WORD(s_h[i], bloom_words) / GF_FSLICE_WORDS], 1u);
  __syncthreads();
  for (int p = tid; p < (int)nparts; p += GF_INDEX_THREADS) {
    const unsigned int c = s_cnt[p];
    s_base[p] = c ? atomicAdd(part_fill + p, c) : 0u;
    s_cnt[p] = 0;
  }
  __syncthreads();
  for (unsigned int i = tid; i < n; i += GF_INDEX_THREADS) {
    const uint32_t h = s_h[i];
    const uint32_t w = GF_BLOOM_WORD(h, bloom_words), p = w / GF_FSLICE_WORDS;
    const unsigned int at = s_base[p] + atomicAdd(&s_cnt[p], 1u);
    if (at < part_cap) {
      part_buf[(size_t)p * part_cap + at] = h;
    } else {  // no room: the old way (the builder ORs over it)
      const uint32_t b = GF_BLOOM_BITS(h);
      if ((__builtin_nontemporal_load(bloom + w) & b) != b) atomicOr(bloom + w, b);
    }
  }
}

// one block per partition: its hashes into 32 KB of LDS, ORed over what the scatter's overflow left in the filter
__global__ __launch_bounds__(1024) void gf_k_filter_build(uint32_t* bloom, uint32_t bloom_words,
                                                          const uint32_t* __restrict__ part_buf, uint32_t part_cap,
                                                          const unsigned int* __restrict__ part_fill) {
  __shared__ unsigned int s_w[GF_FSLICE_WORDS];
  const uint32_t p = blockIdx.x, w0 = p * GF_FSLICE_WORDS;
  for (uint32_t i = threadIdx.x; i < GF_FSLICE_WORDS; i += 1024) s_w[i] = 0;
  __syncthreads();
  const uint32_t n = min(part_fill[p], part_cap);
  const uint32_t* mine = part_buf + (size_t)p * part_cap;
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    const uint32_t h = __builtin_nontemporal_load(mine + i);
    atomicOr(&s_w[GF_BLOOM_WORD(h, bloom_words) - w0], GF_BLOOM_BITS(h));
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < GF_FSLICE_WORDS; i += 1024) {
    const uint32_t w = w0 + i;
    if (w < bloom_words && s_w[i]) bloom[w] |= s_w[i];
  }
}

// Both strands of the genes in site-code space (the even words of gdu, layout: gf_table.h), used by the diagonal
// verification of the mapping kernels.  A thread writes one word = 16 consecutive site codes of one gene's
// region [lin_base - len + 1, lin_base + len - 1]: codes >= lin_base are the forward bases f = code - lin_base,
// codes below it the complements of the bases f = lin_base - code — sixteen bytes of the gene per strand, one
// unaligned load each, converted and (for the reverse strand) turned round.  Bases other than A/C/G/T keep
// code 0, like the gaps between the regions.  (The first form took every base to its two words with atomicOr:
// sixteen neighbouring lanes to the same address, 1.1 ms on a cancer-sized gene set.)
// 16 bases at p (any alignment) as 2-bit fields; valid2 = 3 in the fields of the first `count` bases that are A/C/G/T
__device__ __forceinline__ uint32_t gf_codes16(const uint8_t* p, uint32_t count, uint32_t& valid2) {
  uint4 q;
  __builtin_memcpy(&q, p, 16);
  uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
  gf_convert4(q.x, c0, i0);
  gf_convert4(q.y, c1, i1);
  gf_convert4(q.z, c2, i2);
  gf_convert4(q.w, c3, i3);
  uint32_t m = (i0 | (i1 << 4) | (i2 << 8) | (i3 << 12)) & 0xFFFFu;
  if (count < 16) m |= (0xFFFFu << count) & 0xFFFFu;
  // spread the 16 invalid bits over the 2-bit fields
  m = (m | (m << 8)) & 0x00FF00FFu;
  m = (m | (m << 4)) & 0x0F0F0F0Fu;
  m = (m | (m << 2)) & 0x33333333u;
  m = (m | (m << 1)) & 0x55555555u;
  valid2 = ~(m | (m << 1));
  return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}

__global__ __launch_bounds__(256) void gf_k_index_strands(GfGenes G, const uint32_t* __restrict__ lin_hi,
                                                          uint32_t* __restrict__ gdu, uint32_t gd_words) {
  const uint32_t W = blockIdx.x * blockDim.x + threadIdx.x;
  if (W >= gd_words) return;
  const uint32_t p0 = 16u * W;
  uint32_t out = 0;
  // gene whose slice of the space [lin_hi[c-1], lin_hi[c]) holds p0 (the regions are GF_LIN_PAD apart: a word
  // never holds codes of two genes)
  int lo = -1, hi = G.n_genes;  // invariant lin_hi[lo] <= p0 < lin_hi[hi]  (lin_hi[-1] = 0, lin_hi[n] = inf)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (lin_hi[mid] <= p0) lo = mid; else hi = mid;
  }
  const int c = hi;
  if (c < G.n_genes) {
    const uint32_t base = G.lin_base[c];
    const uint32_t goff = G.gene_off[c];
    const uint32_t len = G.gene_off[c + 1] - goff;
    const uint32_t p1 = p0 + 15u;
    if (len > 0) {
      // forward strand: codes max(p0, base) .. min(p1, base + len - 1)
      if (p1 >= base && p0 <= base + len - 1u) {
        const uint32_t a = p0 > base ? p0 : base, b = p1 < base + len - 1u ? p1 : base + len - 1u;
        uint32_t ok2;
        const uint32_t w = gf_codes16(G.cat + goff + (a - base), b - a + 1u, ok2) & ok2;
        out |= w << (2u * (a - p0));
      }
      // reverse strand: codes max(p0, base - len + 1) .. min(p1, base - 1), f = base - code
      if (len > 1 && p0 + 1u <= base && p1 + len >= base + 1u) {
        const uint32_t a = p0 + len > base + 1u ? p0 : base + 1u - len, b = p1 < base - 1u ? p1 : base - 1u;
        const uint32_t cnt = b - a + 1u, f_lo = base - b;
        uint32_t ok2;
        uint32_t w = gf_codes16(G.cat + goff + f_lo, cnt, ok2);  // field i = base f_lo + i
        w = (w ^ 0xAAAAAAAAu) & ok2;                              // complemented; other bases and the rest: 0
        w = gf_field_reverse(w) >> (2u * (16u - cnt));      // field k = base f_lo + cnt-1-k = base - (a + k)
        out |= w << (2u * (a - p0));
      }
    }
  }
  gdu[2 * (size_t)W] = out;
}

// stats[0]=n_sites [1]=n_keys [2]=n_unique [3]=n_dupe_keys [4]=n_high [5]=n_dupe_sites
// [6]=dupes cursor (the extent of dupes[] handed out, granules' unused ends included)
#define GF_DUPE_GRANULE 512ull
// One sweep over the table after the COUNT pass: counts -> unique / dupes(start in dupes[]) / HIGH, and the
// statistics on the way.  A thread takes one bucket (8 consecutive slots = one 64-byte line), a block 256
// buckets per round: the room in dupes[] for the round's 2..5-fold keys is ONE atomic on the shared counter (a
// block scan hands out the parts).  The first form took one per wavefront: a quarter of a million same-address
// atomics on a cancer-sized table, most of the kernel's 1.16 ms.  The statistics are one atomic per block and
// number (one per wave serialised ~100 K atomics on six addresses).
__global__ __launch_bounds__(256) void gf_k_classify_assign(uint64_t* slots, uint64_t nslots, unsigned long long* stats) {
  // (r03: a lane takes slot PAIRS — one 16-byte load, the wave's loads contiguous — four of them a round; a lane per
  //  bucket read the table in 64-byte strides, 0.41 ms for a cancer-sized table's 467 MB)
  constexpr int NP = 4;
  __shared__ uint32_t s_wave[4];
  __shared__ unsigned long long s_base, s_pool, s_pool_end;
  __shared__ unsigned long long s_part[6][4];
  if (threadIdx.x == 0) s_pool = s_pool_end = 0;  // (read and written by thread 0 alone)
  const uint64_t npairs = nslots / 2;
  const uint64_t per_round = (uint64_t)gridDim.x * blockDim.x * NP;
  const uint64_t rounds = (npairs + per_round - 1) / per_round;  // whole blocks stay in the loop for the barriers
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long sites = 0;
  uint32_t keys = 0, uniq = 0, dk = 0, high = 0, ds = 0;
  for (uint64_t it = 0; it < rounds; ++it) {
    const uint64_t p0 = it * per_round + (uint64_t)blockIdx.x * blockDim.x * NP + threadIdx.x;  // + 256 j
    uint32_t val[2 * NP], want[2 * NP], cnt_of[2 * NP];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const uint64_t p = p0 + 256u * j;
      uint4 q = make_uint4(0, 0, 0, 0);
      if (p < npairs) q = ((const uint4*)slots)[p];
      val[2 * j] = q.x;
      val[2 * j + 1] = q.z;
    }
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
      // a slot holds a count (two-pass build, or a key the one-pass build found more than once), or — one-pass
      // build — the site of a key claimed once: type UNIQUE already, count 1
      const bool claimed = ((val[k] & GF_VAL_LOW) >> GF_TYPE_SHIFT) == GF_TYPE_UNIQUE;
      const uint32_t c = claimed ? 1u : (val[k] & GF_VAL_LOW);
      cnt_of[k] = c;
      want[k] = (c >= 2 && c <= GF_DUP_THRESHOLD) ? c : 0u;
      mine += want[k];
      sites += c;
      keys += c != 0;
      uniq += c == 1;
      dk += want[k] != 0;
      high += c > GF_DUP_THRESHOLD;
    }
    ds += mine;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += s_wave[w];
      total += s_wave[w];
    }
    // Room in dupes[] by the granule: same-address atomics take their turns beyond the L2s, 5-14 ns each, and one
    // per block and round (28 K of them on a cancer-sized table) was most of this kernel's 0.41 ms.  A block takes
    // GF_DUPE_GRANULE entries at a time and hands them out itself; what is left of a granule that the next round
    // does not fit in stays empty (the host sizes dupes[] for it).
    // (A round that needs a quarter of a granule or more and does not fit in what the block holds takes exactly its
    //  own room and leaves the block's granule alone: a gene set of nothing but 2..5-fold keys — the same genes
    //  listed twice — then wastes nothing, and the small rounds of an ordinary one at most a quarter of a granule each.)
    if (threadIdx.x == 0 && total) {
      if (s_pool + total <= s_pool_end) {
        s_base = s_pool;
        s_pool += total;
      } else if (total >= GF_DUPE_GRANULE / 4) {
        s_base = atomicAdd(stats + 6, (unsigned long long)total);
      } else {
        s_pool = atomicAdd(stats + 6, GF_DUPE_GRANULE);
        s_pool_end = s_pool + GF_DUPE_GRANULE;
        s_base = s_pool;
        s_pool += total;
      }
    }
    __syncthreads();
    uint32_t start = (uint32_t)(total ? s_base : 0ull) + before + incl - mine;
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
      const uint32_t c = cnt_of[k];
      if (!c) continue;  // (an empty slot; pairs beyond the table read as empty)
      uint32_t nv;
      if (c == 1) {
        if (((val[k] & GF_VAL_LOW) >> GF_TYPE_SHIFT) == GF_TYPE_UNIQUE) continue;  // claimed with its site: final
        nv = GF_TYPE_UNIQUE << GF_TYPE_SHIFT;
      } else if (c <= GF_DUP_THRESHOLD) {
        nv = (GF_TYPE_DUPES << GF_TYPE_SHIFT) | (c << GF_DUPE_COUNT_SHIFT) | (start & GF_DUPE_START_MASK);
        start += c;
      } else {
        // (the site field: a representative site of the key, the smallest — gf_k_index_side; all ones = none yet)
        nv = (GF_TYPE_HIGH << GF_TYPE_SHIFT) | GF_LIN_MASK;
      }
      *(uint32_t*)(slots + 2 * (p0 + 256u * (k >> 1)) + (k & 1)) = (val[k] & GF_VAL_OVF) | nv;
    }
    __syncthreads();  // s_wave / s_base are rewritten in the next round
  }
  unsigned long long v[6] = {sites, keys, uniq, dk, high, ds};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned long long x = v[k];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if (lane == 0) s_part[k][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    unsigned long long x = 0;
    for (int w = 0; w < 4; ++w) x += s_part[threadIdx.x][w];
    if (x) atomicAdd(stats + threadIdx.x, x);
  }
}

// Sort every duplicate list ascending so its content is reproducible.
__global__ void gf_k_sort_dupes(const uint64_t* slots, uint64_t nslots, uint32_t* dupes) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots;
       s += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t val = (uint32_t)slots[s];
    if (((val & GF_VAL_LOW) >> GF_TYPE_SHIFT) != GF_TYPE_DUPES) continue;
    uint32_t cnt = (val >> GF_DUPE_COUNT_SHIFT) & 7u;
    uint32_t* d = dupes + (val & GF_DUPE_START_MASK);
    for (uint32_t a = 1; a < cnt; ++a) {
      uint32_t x = d[a];
      uint32_t b = a;
      while (b > 0 && d[b - 1] > x) { d[b] = d[b - 1]; --b; }
      d[b] = x;
    }
  }
}

// The strands + flags array once more in overlapping tiles (GfTable::gdt): tile t = pairs 6t .. 6t+15 (zero beyond the
// array's end).  One thread per pair of a tile; runs when the flags are final (after the side list).
#define GF_GDT_STRIDE 6u
__global__ __launch_bounds__(256) void gf_k_gdu_tiles(const uint2* __restrict__ gdu, uint32_t gd_words, uint2* __restrict__ gdt,
                                                      uint32_t n_tiles) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // pair i & 15 of tile i >> 4
  if ((i >> 4) >= n_tiles) return;
  const uint32_t p = GF_GDT_STRIDE * (i >> 4) + (i & 15u);
  gdt[i] = p < gd_words ? gdu[p] : make_uint2(0u, 0u);
}

