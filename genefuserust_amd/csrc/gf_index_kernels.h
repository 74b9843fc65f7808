// K1 — index build on the device.
//
// Reproduces Indexer::make_index + index_contig + fill_bloom_filter
// (src/core/indexer.rs:122-250) as an order-independent classification
// (SURVEY.md Appendix A.1): every valid 16-base window of every gene yields a
// forward site (position f, windows 0..len-17) and a reverse-complement site
// (position -(f+15), windows 1..len-16; the reverse window i of indexer.rs:168
// is the forward window f = len-16-i, so position = i+1-len = -(f+15)).
// Keys seen once keep their site, 2..5 times keep all sites, >= 6 times become
// HIGH.  Three passes over the gene bases, no sort:
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

__device__ __forceinline__ uint64_t gf_atomic_load64(uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// COUNT pass: claim a slot for `key` (or find it) and bump its occurrence count.
__device__ __forceinline__ void gf_insert_count(uint64_t* slots, uint32_t nbuckets, uint32_t key) {
  uint32_t b = gf_bucket_of(key, nbuckets);
  for (uint32_t guard = 0; guard <= nbuckets; ++guard) {
    uint64_t* bucket = slots + (size_t)b * GF_SLOTS_PER_BUCKET;
    for (int j = 0; j < GF_SLOTS_PER_BUCKET; ++j) {
      uint64_t cur = gf_atomic_load64(bucket + j);
      if ((cur & GF_VAL_LOW) == 0) {
        uint64_t want = ((uint64_t)key << 32) | 1ull;
        uint64_t prev = atomicCAS((unsigned long long*)(bucket + j), 0ull, (unsigned long long)want);
        if (prev == 0) return;
        cur = prev;
      }
      if ((uint32_t)(cur >> 32) == key) {
        atomicAdd((unsigned int*)(bucket + j), 1u);  // low word = count
        return;
      }
    }
    atomicOr((unsigned int*)bucket, GF_VAL_OVF);  // slot 0, low word
    b = (b + 1 == nbuckets) ? 0 : b + 1;
  }
}

// Slot holding `key` (must have been inserted).
__device__ __forceinline__ uint64_t* gf_find_slot(uint64_t* slots, uint32_t nbuckets, uint32_t key) {
  uint32_t b = gf_bucket_of(key, nbuckets);
  for (uint32_t guard = 0; guard <= nbuckets; ++guard) {
    uint64_t* bucket = slots + (size_t)b * GF_SLOTS_PER_BUCKET;
    uint64_t first = bucket[0];
    for (int j = 0; j < GF_SLOTS_PER_BUCKET; ++j) {
      uint64_t cur = j ? bucket[j] : first;
      if ((cur & GF_VAL_LOW) != 0 && (uint32_t)(cur >> 32) == key) return bucket + j;
    }
    if (!((uint32_t)first & GF_VAL_OVF)) return nullptr;
    b = (b + 1 == nbuckets) ? 0 : b + 1;
  }
  return nullptr;
}

// FILL pass: store the site code of one occurrence.
// For a key with exactly one site the site's "unique" flag in gdu is set here too (the flag
// word of site code lin is gdu[2 * (lin >> 4) + 1], bit 2 * (lin & 15)): the slot is in hand.
__device__ __forceinline__ void gf_fill_site(uint64_t* slots, uint32_t nbuckets, uint32_t* dupes, uint32_t* gdu,
                                             uint32_t key, uint32_t lin) {
  uint64_t* s = gf_find_slot(slots, nbuckets, key);
  if (!s) return;
  uint32_t* valp = (uint32_t*)s;  // little-endian: low word = val
  uint32_t val = *valp;
  uint32_t type = (val & GF_VAL_LOW) >> GF_TYPE_SHIFT;
  if (type == GF_TYPE_UNIQUE) {
    // exactly one occurrence => exactly one writer
    *valp = (val & GF_VAL_OVF) | (GF_TYPE_UNIQUE << GF_TYPE_SHIFT) | (lin & GF_LIN_MASK);
    atomicOr(gdu + 2 * (lin >> 4) + 1, 1u << (2u * (lin & 15u)));
  } else if (type == GF_TYPE_DUPES) {
    uint32_t cnt = (val >> GF_DUPE_COUNT_SHIFT) & 7u;
    uint32_t start = val & GF_DUPE_START_MASK;
    for (uint32_t k = 0; k < cnt; ++k)
      if (atomicCAS(dupes + start + k, GF_DUPE_EMPTY, lin) == GF_DUPE_EMPTY) break;
  }
}

// One block = one tile of GF_TILE_BASES window starts.  Phase 1 packs the tile's
// ASCII bases (+15 halo) into a 2-bit stream and an invalid-bit stream in LDS with
// coalesced 16-byte loads; phase 2 cuts the windows out of LDS.
template <int MODE>
__global__ __launch_bounds__(GF_INDEX_THREADS) void gf_k_index_sites(GfGenes G, uint64_t* slots,
                                                                     uint32_t nbuckets,
                                                                     uint32_t* dupes, uint32_t* gdu) {
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

  for (int l = tid; l < GF_TILE_BASES; l += GF_INDEX_THREADS) {
    uint32_t g = t0 + (uint32_t)l;
    if (g >= G.total) break;
    uint32_t flags = gf_flags16(s_inv[l >> 5], s_inv[(l >> 5) + 1], (uint32_t)l);
    if (flags) continue;
    // gene of g: last c with gene_off[c] <= g
    int lo = 0, hi = G.n_genes;  // invariant gene_off[lo] <= g < gene_off[hi]
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (G.gene_off[mid] <= g) lo = mid; else hi = mid;
    }
    uint32_t f = g - G.gene_off[lo];
    uint32_t len = G.gene_off[lo + 1] - G.gene_off[lo];
    if (f + GF_KMER > len) continue;  // window runs past the gene
    uint32_t key = gf_window(s_codes[l >> 4], s_codes[(l >> 4) + 1], (uint32_t)l);
    if (f + GF_KMER < len) {  // forward windows 0 .. len-17 (indexer.rs:188)
      if (MODE == GF_MODE_COUNT) gf_insert_count(slots, nbuckets, key);
      else gf_fill_site(slots, nbuckets, dupes, gdu, key, G.lin_base[lo] + f);
    }
    if (f >= 1) {  // reverse windows i = len-16-f in 0 .. len-17
      uint32_t rkey = gf_revcomp_key(key);
      if (MODE == GF_MODE_COUNT) gf_insert_count(slots, nbuckets, rkey);
      else gf_fill_site(slots, nbuckets, dupes, gdu, rkey, G.lin_base[lo] - (f + 15u));
    }
  }
}

// Publish both strands of the genes in site-code space (gdu, layout: gf_table.h), used by the
// diagonal verification of the mapping kernels.  gdu is zero-filled by the host; bits are
// OR-ed in.  (The per-site uniqueness flags of gdu are set by the FILL pass.)
__global__ __launch_bounds__(GF_INDEX_THREADS) void gf_k_index_strands(GfGenes G, uint64_t* slots,
                                                                       uint32_t nbuckets,
                                                                       uint32_t* __restrict__ gdu) {
  __shared__ uint32_t s_codes[GF_TILE_BASES / 16 + 2];
  __shared__ uint32_t s_inv[GF_TILE_BASES / 32 + 2];
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
  }
  __syncthreads();
  for (int l = tid; l < GF_TILE_BASES; l += GF_INDEX_THREADS) {
    const uint32_t g = t0 + (uint32_t)l;
    if (g >= G.total) break;
    int lo = 0, hi = G.n_genes;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (G.gene_off[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t f = g - G.gene_off[lo];
    const uint32_t base = G.lin_base[lo];
    // this base on both strands (invalid bases keep code 0: their windows carry no ub bit)
    const uint32_t code = (s_codes[l >> 4] >> (2 * (l & 15))) & 3u;
    const bool bad = (s_inv[l >> 5] >> (l & 31)) & 1u;
    if (!bad) {
      const uint32_t pf = base + f;
      if (code) atomicOr(gdu + 2 * (pf >> 4), code << (2 * (pf & 15u)));
      if (f >= 1) {
        const uint32_t pr = base - f;  // reverse-complement base j = len-1-f at base + 1 - len + j
        atomicOr(gdu + 2 * (pr >> 4), (code ^ 2u) << (2 * (pr & 15u)));
      }
    }
  }
}

// presence filter over the last and the first 14 bases of every key (gf_table.h: bloom)
__global__ void gf_k_build_bloom(const uint64_t* slots, uint64_t nslots, uint32_t* bloom, uint32_t nwords) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots;
       s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = slots[s];
    if (((uint32_t)v & GF_VAL_LOW) == 0) continue;
    const uint32_t key = (uint32_t)(v >> 32);
    const uint32_t ha = GF_BLOOM_HASH((key >> 4)), hb = GF_BLOOM_HASH((key & 0x0FFFFFFFu));
    // neighbouring keys share their 14-mers, so most bit pairs are set already: look before the atomic
    uint32_t* wa = bloom + GF_BLOOM_WORD(ha, nwords);
    uint32_t* wb = bloom + GF_BLOOM_WORD(hb, nwords);
    const uint32_t ba = GF_BLOOM_BITS(ha), bb = GF_BLOOM_BITS(hb);
    if ((__builtin_nontemporal_load(wa) & ba) != ba) atomicOr(wa, ba);
    if ((__builtin_nontemporal_load(wb) & bb) != bb) atomicOr(wb, bb);
  }
}

// stats[0]=n_sites [1]=n_keys [2]=n_unique [3]=n_dupe_keys [4]=n_high [5]=n_dupe_sites
// [6]=dupes cursor (used by the assign pass)
__global__ void gf_k_classify_count(const uint64_t* slots, uint64_t nslots,
                                    unsigned long long* stats) {
  unsigned long long sites = 0, keys = 0, uniq = 0, dk = 0, high = 0, ds = 0;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nslots;
       s += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t c = (uint32_t)slots[s] & GF_VAL_LOW;
    if (!c) continue;
    sites += c;
    keys += 1;
    if (c == 1) uniq += 1;
    else if (c <= GF_DUP_THRESHOLD) { dk += 1; ds += c; }
    else high += 1;
  }
  // one atomic per block and statistic (one per wave serialised ~100 K atomics on six addresses)
  __shared__ unsigned long long s_part[6][4];
  unsigned long long v[6] = {sites, keys, uniq, dk, high, ds};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned long long x = v[k];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if ((threadIdx.x & 63) == 0) s_part[k][(threadIdx.x >> 6) & 3] = x;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    unsigned long long x = 0;
    for (unsigned int w = 0; w < (blockDim.x + 63) / 64 && w < 4; ++w) x += s_part[threadIdx.x][w];
    if (x) atomicAdd(stats + threadIdx.x, x);
  }
}

// A thread takes one bucket (8 consecutive slots = one 64-byte line), a block 256 buckets per round: the room
// in dupes[] for the round's 2..5-fold keys is ONE atomic on the shared counter (a block scan hands out the
// parts).  The first form took one per wavefront: a quarter of a million same-address atomics on a
// cancer-sized table, most of the kernel's 1.16 ms.
__global__ __launch_bounds__(256) void gf_k_classify_assign(uint64_t* slots, uint64_t nslots, unsigned long long* stats) {
  __shared__ uint32_t s_wave[4];
  __shared__ unsigned long long s_base;
  const uint64_t nbuckets = nslots / GF_SLOTS_PER_BUCKET;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t rounds = (nbuckets + stride - 1) / stride;  // whole blocks stay in the loop for the barriers
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t it = 0; it < rounds; ++it) {
    const uint64_t b = it * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t val[GF_SLOTS_PER_BUCKET], want[GF_SLOTS_PER_BUCKET];
    uint32_t mine = 0;
    uint64_t* base_slot = slots + b * GF_SLOTS_PER_BUCKET;
#pragma unroll
    for (int k = 0; k < GF_SLOTS_PER_BUCKET; ++k) {
      val[k] = b < nbuckets ? (uint32_t)base_slot[k] : 0u;
      const uint32_t c = val[k] & GF_VAL_LOW;
      want[k] = (c >= 2 && c <= GF_DUP_THRESHOLD) ? c : 0u;
      mine += want[k];
    }
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
    if (threadIdx.x == 0 && total) s_base = atomicAdd(stats + 6, (unsigned long long)total);
    __syncthreads();
    uint32_t start = (uint32_t)(total ? s_base : 0ull) + before + incl - mine;
    if (b < nbuckets) {
#pragma unroll
      for (int k = 0; k < GF_SLOTS_PER_BUCKET; ++k) {
        const uint32_t c = val[k] & GF_VAL_LOW;
        if (!c) continue;
        uint32_t nv;
        if (c == 1) {
          nv = GF_TYPE_UNIQUE << GF_TYPE_SHIFT;
        } else if (c <= GF_DUP_THRESHOLD) {
          nv = (GF_TYPE_DUPES << GF_TYPE_SHIFT) | (c << GF_DUPE_COUNT_SHIFT) | start;
          start += c;
        } else {
          nv = GF_TYPE_HIGH << GF_TYPE_SHIFT;
        }
        *(uint32_t*)(base_slot + k) = (val[k] & GF_VAL_OVF) | nv;
      }
    }
    __syncthreads();  // s_wave / s_base are rewritten in the next round
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
