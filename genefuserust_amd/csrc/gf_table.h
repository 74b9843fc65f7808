// Device table format shared by the index-build and mapping kernels.
//
// Reference semantics reproduced (src/core/indexer.rs:179-250): k-mer ->
//   unique site | 2..5 sites | HIGH (>=6, no sites) | absent.
// The reference pairs an exact 2^32-bit bitmap with an FxHashMap; here one
// probe of a 64-byte bucket answers both "present?" and "where?".
//
// Layout (all in HBM, sized to stay resident in the 256 MiB Infinity Cache for
// druggable-sized panels):
//   slots[nbuckets][8]  : uint64 = (key << 32) | val
//   dupes[]             : uint32 site codes of the 2..5-fold keys
//   val  bit 31         : (slot 0 of a bucket only) some key that hashes to this
//                         bucket was placed in a later bucket -> keep probing
//        bits 30..29    : 0 with low bits 0 = empty, 1 = unique, 2 = dupes, 3 = HIGH
//        unique         : bits 28..0 = lin (site code)
//        dupes          : bits 28..26 = count (2..5), bits 25..0 = first index in dupes[]
//
// key  = 16 bases at 2 bits each, base j of the window in bits [2j, 2j+1],
//        code = (ascii >> 1) & 3  (A=0 C=1 T=2 G=3).  This equals the bit
//        reversal of the reference's k-mer (indexer.rs:789-913: first base most
//        significant, A=0 T=1 C=2 G=3), a bijection, so hit/miss decisions are
//        exact; k-mer values never appear in any output.
// lin  = lin_base[contig] + position (position < 0 on the reverse strand), a
//        32-bit code of GenePos.  Contig c owns the interval
//        [lin_base[c] - len_c - GF_LIN_PAD, lin_base[c] + len_c) so that
//        lin - i (the vote "diagonal" GenePos{contig, position - i},
//        indexer.rs:690-706) stays inside contig c's interval for every read
//        offset i < GF_LIN_PAD and two different GenePos never share a code.
#pragma once

#include <stdint.h>

#define GF_KMER 16
#define GF_SLOTS_PER_BUCKET 8
#define GF_LIN_PAD 4096u /* >= GF_MAX_READ_LEN */

#define GF_VAL_OVF 0x80000000u
#define GF_VAL_LOW 0x7FFFFFFFu
#define GF_TYPE_SHIFT 29
#define GF_TYPE_UNIQUE 1u
#define GF_TYPE_DUPES 2u
#define GF_TYPE_HIGH 3u
#define GF_LIN_MASK 0x1FFFFFFFu
#define GF_DUPE_COUNT_SHIFT 26
#define GF_DUPE_START_MASK 0x03FFFFFFu
#define GF_DUPE_EMPTY 0xFFFFFFFFu

// thresholds: src/aux/global_settings.rs:23-26, src/core/indexer.rs:619-620
#define GF_DUP_THRESHOLD 5
#define GF_MAJOR_KEYS 40
#define GF_MINOR_KEYS 20
#define GF_MISMATCH_THRESHOLD 10
#define GF_ALLOWED_GAP 10
#define GF_THRESHOLD_LEN 20

struct GfTable {
  const uint64_t* slots;    // nbuckets * 8
  const uint32_t* dupes;    // duplicate-site lists
  const uint32_t* lin_base; // [n_genes]   lin of (contig, position 0)
  const uint32_t* lin_hi;   // [n_genes]   exclusive upper end of contig's interval
  const uint32_t* gene_len; // [n_genes]
  // diagonal verification (gf_map_kernels.h, seed+verify): both strands of every gene
  // laid out in site-code ("lin") space, so that the 16 bases of the site with code
  // lin sit at positions lin .. lin+15 whatever its contig and strand:
  //   forward base f of contig c            at lin_base[c] + f
  //   reverse-complement base j (0-based)   at lin_base[c] + 1 - len_c + j
  // (the two meet at lin_base[c]; it is given to the forward strand — the reverse
  // strand's last base belongs to no indexed window, indexer.rs:188.)
  // gdu[2k] = 16 bases (2 bits each) at positions 16k..16k+15; gdu[2k+1] = flags in the
  // same layout, bit 2(p%16): the site with code p exists and is the only site of its
  // key.  Interleaved so that one cache line serves both.
  const uint32_t* gdu;
  uint32_t gd_words;        // word pairs in gdu: positions 0 .. 16 * gd_words - 1 are addressable
  // The same pairs once more in overlapping 128-byte tiles (r04; nullptr = not built): tile t = pairs 6t .. 6t+15,
  // so the 11 pairs a 150-base read's diagonal needs (seed+verify, reads of up to 160 bases) starting at ANY pair p
  // lie in ONE cache line — tile p / 6, from its pair p % 6 on — where the plain array's 88 bytes straddle two lines
  // for 10 of the 16 starting positions (1.7 missed lines per on-target read).  2.67 x the bytes.
  const uint32_t* gdt;
  // presence filter over 14-mers (one 32-bit word, two bits per element), consulted
  // before any bucket probe of a window that is expected to miss.  For every key of the
  // table its last 14 bases (key >> 4) and its first 14 bases (key & 0x0FFFFFFF) are
  // inserted.  Two consecutive stride-2 windows w, w+1 share the 14-mer S = bases 2..15 of
  // w = bases 0..13 of w+1, so ONE lookup of S answers for both: S absent => neither
  // window's key is in the table (no false negatives).  Sized to stay resident in the
  // 4 MiB L2 of each XCD, where a lookup costs a fraction of an L2-missing bucket probe.
  const uint32_t* bloom;
  uint32_t bloom_words;     // 0 = filter disabled
  uint32_t bloom_in_l2;     // 1 = (mostly) L2 hits: also worth asking for the seeds; 2 = fits an XCD's L2:
                            //     seed+verify runs the filter pass for reads without a candidate itself
  uint32_t nbuckets;
  int32_t n_genes;
  // per-call, optional: read r is not searched (its result is []) when skip[r] > 0.  The pair
  // pipeline maps R1 / R2 in place and skips the pairs whose merged read is searched instead
  // (pescanner.rs:446-471): skip = the merged lengths.
  const int32_t* skip;
  // per-call, optional: > 0 = every read has exactly this many bases and read r starts at base r * fixed_len: the
  // kernels compute the offsets instead of loading them (gf_map_reads_fixed_device; r03: the int64 offsets are 2.5 M of
  // seed+verify's 75.6 M missed lines per 20 M reads: -2.3 % of its time)
  int32_t fixed_len;
  // per-call, optional: a count on the device — only the reads below it exist (the others are empty slots at the end
  // of the batch: gf_scan_pairs_device's merged reads, compacted to the front of a batch of n slots whose number the
  // host does not know).  Seed+verify then does not walk the empty slots (r03 b: 0.3 of 0.55 ms per 10 M pairs).
  const int64_t* n_dev;
  // per-call, optional (the wave-per-read kernels of a zero-copy host call, gfmatch.hip zc_submit).  done_flag set:
  // counts / matches are pinned host memory whose counts start out zero — only reads with segments are written, each
  // followed by a system-scope fence of its wave.  done_ctr set as well (the call's last launch): every block counts
  // itself out on it, and the last one stores done_seq to done_flag — the host waits on that word instead of a
  // stream synchronisation.
  unsigned int* done_ctr;
  unsigned int* done_flag;
  uint32_t done_seq;
};

// filter word and bit pair of a 14-mer x (28 bits).  One multiplicative hash: the word comes
// from its high bits (mulhi), the two bit positions from its low bits folded with the
// middle ones.  (A full murmur finaliser here cost four 32-bit multiplies per look-up,
// quarter-rate instructions, in a kernel that does 34 look-ups per read.)
// Both strands of every gene are indexed, so nearly every 14-mer in the table is there with
// its reverse complement: the filter stores the smaller of the two (and looks up the smaller
// of the two), which halves the items it has to hold — at 3 MiB that is 6 bits per item
// instead of 3, a false-positive rate of ~8 % instead of ~22 %.  No symmetry is *assumed*:
// every key's two 14-mers are inserted in canonical form, whatever else the table holds.
#define GF_BLOOM_HASH(x) ((uint32_t)gf_canon14(x) * 0x9E3779B1u)
#define GF_BLOOM_BITS(h) ((1u << (((h) ^ ((h) >> 15)) & 31u)) | (1u << ((((h) ^ ((h) >> 15)) >> 5) & 31u)))
#define GF_BLOOM_WORD(h, nwords) ((uint32_t)(((uint64_t)(h) * (uint64_t)(nwords)) >> 32))

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

// murmur3 finaliser: a bijection of 32-bit words, good avalanche.
GF_HD uint32_t gf_mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

GF_HD uint32_t gf_bucket_of(uint32_t key, uint32_t nbuckets) {
  return (uint32_t)(((uint64_t)gf_mix32(key) * (uint64_t)nbuckets) >> 32);
}

// reverse the order of the 16 2-bit fields
GF_HD uint32_t gf_field_reverse(uint32_t x) {
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
}

// key of the reverse-complement window (complement = code ^ 2 under A0 C1 T2 G3)
GF_HD uint32_t gf_revcomp_key(uint32_t key) { return gf_field_reverse(key) ^ 0xAAAAAAAAu; }

// canonical form of a 14-mer (28 bits, base 0 in the low bits): the smaller of it and its
// reverse complement
GF_HD uint32_t gf_canon14(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r = __brev(x);  // fields reversed, and the two bits of each field swapped
  r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
#else
  uint32_t r = gf_field_reverse(x);
#endif
  r = (r >> 4) ^ 0x0AAAAAAAu;
  return r < x ? r : x;
}

// reference-coded k-mer (indexer.rs:789-913) -> device key
GF_HD uint32_t gf_key_from_ref_kmer(uint32_t k) {
  k = ((k >> 1) & 0x55555555u) | ((k & 0x55555555u) << 1);
  k = ((k >> 2) & 0x33333333u) | ((k & 0x33333333u) << 2);
  k = ((k >> 4) & 0x0F0F0F0Fu) | ((k & 0x0F0F0F0Fu) << 4);
  k = ((k >> 8) & 0x00FF00FFu) | ((k & 0x00FF00FFu) << 8);
  return (k >> 16) | (k << 16);
}

// ---- how many first-pass votes ONE diagonal can still collect (r03) ----
// indexer.rs:275-321 gives a diagonal one vote per stride-2 window whose key has a site on it.  In site-code
// space (gdu, above) a vote of window w (read base 2w) for diagonal K says: read[2w .. 2w+16) == G[K+2w .. K+2w+16)
// and K+2w is an indexed site.  Two windows u < v that vote for the same K with v - u <= 7 (their bases overlap
// or touch with at most 14 between the starts) lie on ONE strand of one gene — a reverse-strand site and a
// forward site of a contig are at least 16 codes apart, contigs 4096 — so read[2u .. 2v+16) equals G there and
// every site code between K+2u and K+2v is an indexed site with exactly the read's window on it: the key of
// EVERY window between u and v is in the table (indexer.rs:188-240 files a site for every valid window of a
// strand but the last).  Contrapositive, which is what the presence filter can use: two voters of one diagonal
// with a window between them whose key is provably absent are at least 8 windows apart.  So with `can` = the
// windows not yet ruled out, a diagonal's voters V satisfy: u, v in V, v - u <= 7  =>  all of u..v in `can`;
// the largest such V is an upper bound of count1 (indexer.rs:336-346), far below popcount(can) when the
// windows ruled out are spread evenly: 2 of every 4 windows ruled out leave at most 2 votes per 12 windows.
//
// Pairs: the filter answers for windows 2P, 2P+1 together (they share a 14-mer), so the bound is computed on
// pairs — pair P counts as standing when either of its windows does (more windows can only raise the bound).
// h[P] = most voters of a set whose last voter is window 2P+1, F[P] = max h[0..P]:
//   h[P] = 2 + max(h[P-1], F[P-5], h[P-4] - 1)   (P standing; else 0)
// (h[P-1]: the same run of standing windows; F[P-5]: a voter at or below window 2P-9, 8 or more below window 2P;
//  h[P-4] - 1: window 2P-8 last, i.e. pair P-4 without its upper window.)  tests/test_vote_bound.py checks the
// recurrence against exhaustive search and the statement itself against the oracle's vote lists.
// x[k] bit 2q = pair 16k + q standing (bits at the even positions, as (w | w >> 1) & 0x55555555 leaves them).
template <int NPAIRS>
GF_HD int gf_vote_bound_pairs(const uint32_t* x) {
  int h1 = 0, h2 = 0, h3 = 0, h4 = 0;          // h[P-1] .. h[P-4]
  int f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;  // F[P-1] .. F[P-5]
#pragma unroll
  for (int P = 0; P < NPAIRS; ++P) {
    const int standing = -(int)((x[P >> 4] >> (2 * (P & 15))) & 1u);  // 0 or ~0
    int t = h1 > f5 ? h1 : f5;
    t = t > h4 - 1 ? t : h4 - 1;
    const int h = (t + 2) & standing;
    const int f = f1 > h ? f1 : h;
    h4 = h3; h3 = h2; h2 = h1; h1 = h;
    f5 = f4; f4 = f3; f3 = f2; f2 = f1; f1 = f;
  }
  return f1;
}

// 4 ASCII bases in one little-endian dword -> 8 bits of codes (first base in the
// low bits) and 4 invalid flags.  Valid bases are exactly 'A','C','G','T'
// (indexer.rs:825-841: anything else, lower case included, voids the window).
GF_HD void gf_convert4(uint32_t x, uint32_t& code8, uint32_t& inv4) {
  uint32_t y = (x >> 1) & 0x03030303u;
  code8 = (y * 0x01041040u) >> 24;
  // bit (b & 31) of 0x0010008A is set for b in {0x41,0x43,0x47,0x54} given (b & 0xE0) == 0x40
  uint32_t inv = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t b = (x >> (8 * j)) & 0xFFu;
    uint32_t ok = (((b & 0xE0u) == 0x40u) ? 1u : 0u) & (0x0010008Au >> (b & 31u));
    inv |= (ok ^ 1u) << j;
  }
  inv4 = inv;
}

// 32-bit window starting at base g of a little-endian 2-bit stream held in dwords.
GF_HD uint32_t gf_window(uint32_t lo, uint32_t hi, uint32_t g) {
  uint32_t sh = (g & 15u) * 2u;
  return sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
}

// 16 flag bits starting at bit g of a little-endian bit stream held in dwords.
GF_HD uint32_t gf_flags16(uint32_t lo, uint32_t hi, uint32_t g) {
  uint32_t sh = g & 31u;
  uint32_t v = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
  return v & 0xFFFFu;
}
