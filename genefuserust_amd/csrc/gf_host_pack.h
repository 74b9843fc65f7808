// Host side of the packed hand-over (include/gfmatch.h: gf_pack_bases_host): ASCII bases -> the form the PACKED
// kernels take, bit for bit what gf_k_pack_bases writes (gf_pipe_kernels.h): chunk c of pk = bases 16c .. 16c+15 at
// 2 bits each, code = (ascii >> 1) & 3 whatever the byte is; bit j of iv[c] = base j is not one of A C G T
// (indexer.rs:825-841: anything else voids the window); bases at or beyond n_bases are zero bytes and bad.
// A host that packs its reads ships 6 bytes per 16 bases over the link instead of 16.
// AVX2 where the CPU has it (32 bases per step: one table look-up gives the expected letters, two multiply-adds
// fold four 2-bit codes into a byte), plain C++ otherwise; both produce the same words.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

static inline void gf_host_pack_chunk_scalar(const unsigned char* b, int64_t avail, uint32_t* pk, uint16_t* iv) {
  uint32_t code = 0, bad = 0;
  for (int j = 0; j < 16; ++j) {
    const unsigned char ch = j < avail ? b[j] : 0;
    code |= (uint32_t)((ch >> 1) & 3u) << (2 * j);
    const bool ok = j < avail && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
    bad |= (ok ? 0u : 1u) << j;
  }
  *pk = code;
  *iv = (uint16_t)bad;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static void gf_host_pack_avx2(const unsigned char* bases, int64_t c0, int64_t c1,
                                                             uint32_t* pk, uint16_t* iv) {
  // chunks c0 .. c1-1, all of them whole (32 readable bytes from 16*c for every pair of chunks handled here)
  const __m256i three = _mm256_set1_epi8(3);
  const __m256i letters = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i w14 = _mm256_set1_epi16(0x0401);        // bytes (1, 4): b0 + 4 b1
  const __m256i w116 = _mm256_set1_epi32(0x00100001);   // words (1, 16): + 16 (b2 + 4 b3)
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                        0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  int64_t c = c0;
  for (; c + 2 <= c1; c += 2) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)(bases + 16 * c));
    const __m256i y = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
    const __m256i e = _mm256_shuffle_epi8(letters, y);
    const uint32_t good = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(e, v));
    const __m256i p = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(y, w14), w116), pick);
    pk[c] = (uint32_t)_mm256_extract_epi32(p, 0);
    pk[c + 1] = (uint32_t)_mm256_extract_epi32(p, 4);
    iv[c] = (uint16_t)~good;
    iv[c + 1] = (uint16_t)(~good >> 16);
  }
  for (; c < c1; ++c) gf_host_pack_chunk_scalar(bases + 16 * c, 16, pk + c, iv + c);
}
#endif

// chunks [c0, c1) of the stream; n_bases = bases that exist (chunks at or past the end are "all bad")
static inline void gf_host_pack_range(const unsigned char* bases, int64_t n_bases, int64_t c0, int64_t c1, uint32_t* pk,
                                      uint16_t* iv) {
  const int64_t whole = n_bases / 16;  // chunks with all 16 bases present
  int64_t c = c0;
#if defined(__x86_64__)
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  if (have_avx2 && c < whole && c < c1) {
    const int64_t e = whole < c1 ? whole : c1;
    gf_host_pack_avx2(bases, c, e, pk, iv);
    c = e;
  }
#endif
  for (; c < c1; ++c) {
    const int64_t avail = n_bases - 16 * c;
    gf_host_pack_chunk_scalar(bases + (avail > 0 ? 16 * c : 0), avail > 0 ? (avail < 16 ? avail : 16) : 0, pk + c, iv + c);
  }
}
