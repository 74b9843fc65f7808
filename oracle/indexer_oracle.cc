// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the GeneFuseRust `Indexer` hot path, written from the
// reference's Rust source (read as text; the Rust toolchain is not available in
// this pipeline, so the reference itself can never be run here).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
// shared object.  The product library (genefuserust_amd/csrc) never links,
// loads or calls anything in oracle/.
//
// "Parity unpinned": the reference's own tests hold no expected value for
// make_index / map_read / segment_mask, so this oracle is
// pinned only by (a) hand-derived known answers (SURVEY.md Appendix B,
// tests/test_oracle_kat.py) and (b) agreement with a second, independently
// written model (oracle/indexer_model.py).  One function here IS pinned by the
// reference: orc_fast_merge reproduces the expected merged read of the reference's
// unit test (read.rs:450-486; tests/golden/fast_merge_ref_test.json).
//
// The code is deliberately naive and keeps the reference's structure: an exact
// 2^32-bit membership bitmap ("bloom filter"), a hash map k-mer -> GenePos with
// side lists for 2..5-fold duplicates and a HIGH marker for >=6, an ordered
// map for the per-read votes, one read per call.
//
// Each function cites the reference lines it restates (paths relative to
// /root/reference).

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// src/core/indexer.rs:30-38
constexpr uint8_t MATCH_TOP = 3;
constexpr uint8_t MATCH_SECOND = 2;
constexpr uint8_t MATCH_NONE = 1;
constexpr uint8_t MATCH_UNKNOWN = 0;
constexpr int32_t KMER = 16;
constexpr size_t BLOOM_FILTER_SIZE = size_t(1) << 29;  // bytes => 2^32 bits

// src/core/common.rs:31-32
constexpr int16_t DUPE_NORMAL_LEVEL = -1;
constexpr int16_t DUPE_HIGH_LEVEL = -2;

// src/aux/global_settings.rs:23-26 (never changed at run time: genefuse.rs:47-50)
constexpr size_t SKIP_KEY_DUP_THRESHOLD = 5;
constexpr int32_t MAJOR_GENE_KEY_REQUIREMENT = 40;
constexpr int32_t MINOR_GENE_KEY_REQUIREMENT = 20;
constexpr int32_t MISMATCH_THRESHOLD = 10;

// src/core/common.rs:4-7
struct GenePos {
  int16_t contig;
  int32_t position;
};

// src/core/indexer.rs:41-45
struct SeqMatch {
  int32_t seq_start;
  int32_t seq_end;
  GenePos start_gp;
};

// src/core/sequence.rs:51-59
inline char complement_base(char b) {
  switch (b) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

// src/core/sequence.rs:22-50
std::string reverse_complement(const std::string& s) {
  std::string out(s.size(), 'N');
  for (size_t i = 0; i < s.size(); ++i) out[s.size() - 1 - i] = complement_base(s[i]);
  return out;
}

// src/core/indexer.rs:789-850 (make_kmer_cv) and :852-913 (make_kmer_bytes);
// the two differ only in the container type of `seq`.
int64_t make_kmer(const char* seq, int32_t pos, int64_t last_kmer, int32_t step) {
  int64_t kmer = 0;
  int32_t start = 0;
  if (last_kmer >= 0) {
    kmer = last_kmer;
    start = KMER - step;
    if (step == 1) kmer = (kmer & 0x3FFFFFFF) << 2;
    else if (step == 2) kmer = (kmer & 0x0FFFFFFF) << 2;
    else if (step == 3) kmer = (kmer & 0x03FFFFFF) << 2;
    else if (step == 4) kmer = (kmer & 0x00FFFFFF) << 2;
  }
  for (int32_t i = start; i < KMER; ++i) {
    switch (seq[pos + i]) {
      case 'A': kmer += 0; break;
      case 'T': kmer += 1; break;
      case 'C': kmer += 2; break;
      case 'G': kmer += 3; break;
      default: return -1;
    }
    if (i < KMER - 1) kmer <<= 2;
  }
  return kmer;
}

// src/core/indexer.rs:690-695
inline GenePos shift(const GenePos& gp, int32_t i) { return GenePos{gp.contig, gp.position - i}; }

// src/core/indexer.rs:698-706 (little-endian: low word = position bits, high word = 0,
// then OR with the sign-extended contig shifted left by 32)
inline int64_t gp_to_i64(const GenePos& gp) {
  int64_t ret = (int64_t)gp.contig;
  int64_t low = (int64_t)(uint64_t)(uint32_t)gp.position;
  return (int64_t)((uint64_t)ret << 32) | low;
}

// src/core/indexer.rs:709-714
inline GenePos i64_to_gp(int64_t v) {
  return GenePos{(int16_t)(v >> 32), (int32_t)(uint32_t)(v & 0x00000000FFFFFFFFLL)};
}

// src/core/indexer.rs:716-732
void make_mask(uint8_t* mask, uint8_t flag, int32_t seqlen, int32_t start, int32_t kmer_size) {
  int32_t end_point = std::min(seqlen, start + kmer_size);
  for (int32_t p = start; p < end_point; ++p) mask[p] = std::max(mask[p], flag);
}

// src/core/indexer.rs:616-679
std::vector<SeqMatch> segment_mask(const uint8_t* mask, int32_t seqlen, GenePos gp1, GenePos gp2) {
  std::vector<SeqMatch> result;
  const int32_t ALLOWED_GAP = 10;
  const int32_t THRESHOLD_LEN = 20;
  const int32_t targets[2] = {MATCH_TOP, MATCH_SECOND};
  const GenePos gps[2] = {gp1, gp2};
  for (int t = 0; t < 2; ++t) {
    const int32_t target = targets[t];
    int32_t max_start = -1, max_end = -1;
    int32_t start = 0, end = 0;
    while (true) {
      while ((int32_t)mask[start] != target && start != seqlen - 1) start += 1;
      if (start >= seqlen - 1) break;
      if ((int32_t)mask[start] == target) {
        end = start + 1;
        int32_t g = 0;
        while (g < ALLOWED_GAP && (end + g) < seqlen) {
          if ((int32_t)mask[end + g] > target) break;
          if (end + g < seqlen && (int32_t)mask[end + g] == target) {
            end += g + 1;
            g = 0;
            continue;
          }
          g += 1;
        }
        end -= 1;
        if (end - start > (max_end - max_start)) {
          max_end = end;
          max_start = start;
        }
        start += 1;
      } else {
        break;
      }
    }
    if (max_end - max_start > THRESHOLD_LEN) result.push_back(SeqMatch{max_start, max_end, gps[t]});
  }
  return result;
}

struct Indexer {
  // src/core/indexer.rs:67-78
  int32_t m_unique_pos = 0;
  int32_t m_dupe_pos = 0;
  std::unordered_map<int64_t, GenePos> m_kmer_pos;
  uint8_t* m_bloom_filter = nullptr;  // calloc: untouched pages stay unmapped
  std::vector<std::vector<GenePos>> m_dupe_list;
  std::vector<std::string> m_fusion_seq;

  Indexer() { m_bloom_filter = (uint8_t*)calloc(BLOOM_FILTER_SIZE, 1); }
  ~Indexer() { free(m_bloom_filter); }

  // src/core/indexer.rs:179-241
  void index_contig(size_t ctg, const std::string& seq, int32_t start) {
    int64_t kmer = -1;
    for (int32_t i = 0; i < (int32_t)seq.size() - KMER; ++i) {
      kmer = make_kmer(seq.data(), i, kmer, 1);
      if (kmer < 0) continue;
      GenePos site{(int16_t)ctg, i + start};
      auto it = m_kmer_pos.find(kmer);
      if (it != m_kmer_pos.end()) {
        GenePos gp = it->second;
        if (gp.contig == DUPE_HIGH_LEVEL) {
          continue;
        } else if (gp.contig == DUPE_NORMAL_LEVEL) {
          if (m_dupe_list[gp.position].size() >= SKIP_KEY_DUP_THRESHOLD) {
            it->second.contig = DUPE_HIGH_LEVEL;
            m_dupe_list[gp.position] = std::vector<GenePos>();
          } else {
            m_dupe_list[gp.position].push_back(site);
          }
        } else {
          std::vector<GenePos> gps;
          gps.push_back(gp);
          gps.push_back(site);
          m_dupe_list.push_back(gps);
          it->second.contig = DUPE_NORMAL_LEVEL;
          it->second.position = (int32_t)(m_dupe_list.size() - 1);
          m_unique_pos -= 1;
          m_dupe_pos += 1;
        }
      } else {
        m_kmer_pos.emplace(kmer, site);
        m_unique_pos += 1;
      }
    }
  }

  // src/core/indexer.rs:243-250
  void fill_bloom_filter() {
    for (const auto& kv : m_kmer_pos) {
      m_bloom_filter[(size_t)(kv.first >> 3)] |= (uint8_t)(1u << (kv.first & 0x07));
    }
  }

  // src/core/indexer.rs:122-177, from the point where the gene slice has been cut
  // out of the chromosome (:154-158).  `raw` is contig[m_start..m_end] as bytes;
  // a null/absent gene (chromosome not found, :149-150) is passed as len < 0.
  void make_index(const char* const* raw, const int64_t* lens, int32_t n_genes) {
    for (int32_t ctg = 0; ctg < n_genes; ++ctg) {
      if (lens[ctg] < 0) {
        m_fusion_seq.push_back("");
        continue;
      }
      std::string s(raw[ctg], (size_t)lens[ctg]);
      for (auto& ch : s)  // :159 to_uppercase (ASCII genome)
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
      index_contig((size_t)ctg, s, 0);
      std::string rc = reverse_complement(s);
      index_contig((size_t)ctg, rc, 1 - (int32_t)s.size());
      m_fusion_seq.push_back(s);
    }
    fill_bloom_filter();
  }

  // src/core/indexer.rs:252-538
  std::vector<SeqMatch> map_read(const char* seq, int32_t seqlen) const {
    std::map<int64_t, int32_t> kmer_stat;
    kmer_stat[0] = 0;
    const int32_t step = 2;

    // first pass (:275-321)
    int64_t kmer = -1;
    for (int32_t i = 0; i < seqlen - KMER + 1; i += step) {
      kmer = make_kmer(seq, i, kmer, step);
      if (kmer < 0) continue;
      int64_t pos = kmer >> 3;
      int64_t bit = kmer & 0x07;
      if ((m_bloom_filter[pos] & (uint8_t)(1u << bit)) == 0) {
        kmer_stat[0] += 1;
        continue;
      }
      const GenePos& gp = m_kmer_pos.at(kmer);
      if (gp.contig == DUPE_HIGH_LEVEL) {
        continue;
      } else if (gp.contig == DUPE_NORMAL_LEVEL) {
        const auto& lst = m_dupe_list[gp.position];
        for (size_t g = 0; g < lst.size(); ++g) kmer_stat[gp_to_i64(shift(lst[g], i))] += 1;
      } else {
        kmer_stat[gp_to_i64(shift(gp, i))] += 1;
      }
    }

    // top two (:323-346)
    int64_t gp1 = 0, gp2 = 0;
    int32_t count1 = 0, count2 = 0;
    for (const auto& kv : kmer_stat) {
      if (kv.first != 0 && kv.second > count1) {
        gp2 = gp1;
        count2 = count1;
        gp1 = kv.first;
        count1 = kv.second;
      } else if (kv.first != 0 && kv.second > count2) {
        gp2 = kv.first;
        count2 = kv.second;
      }
    }

    // gate (:353-360)
    if (count1 * step < MAJOR_GENE_KEY_REQUIREMENT || count2 * step < MINOR_GENE_KEY_REQUIREMENT)
      return {};

    // second pass (:362-521)
    std::vector<uint8_t> mask((size_t)seqlen, MATCH_UNKNOWN);
    kmer = -1;
    for (int32_t i = 0; i < seqlen - KMER + 1; ++i) {
      kmer = make_kmer(seq, i, kmer, 1);
      if (kmer < 0) continue;
      int64_t pos = kmer >> 3;
      int64_t bit = kmer & 0x07;
      if ((m_bloom_filter[pos] & (uint8_t)(1u << bit)) == 0) continue;
      const GenePos& gp = m_kmer_pos.at(kmer);
      if (gp.contig == DUPE_HIGH_LEVEL) {
        continue;
      } else if (gp.contig == DUPE_NORMAL_LEVEL) {
        const auto& lst = m_dupe_list[gp.position];
        for (size_t g = 0; g < lst.size(); ++g) {
          int64_t gplong = gp_to_i64(shift(lst[g], i));
          if (std::llabs(gplong - gp1) <= 1) make_mask(mask.data(), MATCH_TOP, seqlen, i, KMER);
          else if (std::llabs(gplong - gp2) <= 1) make_mask(mask.data(), MATCH_SECOND, seqlen, i, KMER);
          else if (gplong == 0) make_mask(mask.data(), MATCH_NONE, seqlen, i, KMER);
        }
      } else {
        int64_t gplong = gp_to_i64(shift(gp, i));
        if (std::llabs(gplong - gp1) <= 1) make_mask(mask.data(), MATCH_TOP, seqlen, i, KMER);
        else if (std::llabs(gplong - gp2) <= 1) make_mask(mask.data(), MATCH_SECOND, seqlen, i, KMER);
        else if (gplong == 0) make_mask(mask.data(), MATCH_NONE, seqlen, i, KMER);
      }
    }

    // mismatch gate (:523-535)
    int32_t mismatches = 0;
    for (int32_t p = 0; p < seqlen; ++p)
      if (mask[p] == MATCH_NONE || mask[p] == MATCH_UNKNOWN) mismatches += 1;
    if (mismatches > MISMATCH_THRESHOLD) return {};

    return segment_mask(mask.data(), seqlen, i64_to_gp(gp1), i64_to_gp(gp2));
  }
};

}  // namespace

// ---------------------------------------------------------------------------
// C entry points (ctypes).  Plain pointers and sizes only.
// ---------------------------------------------------------------------------
extern "C" {

struct orc_seqmatch {
  int32_t seq_start;
  int32_t seq_end;
  int32_t position;
  int16_t contig;
  int16_t pad;
};

void* orc_index_build(const char* const* gene_seqs, const int64_t* gene_lens, int32_t n_genes) {
  Indexer* ix = new Indexer();
  ix->make_index(gene_seqs, gene_lens, n_genes);
  return ix;
}

void orc_index_free(void* h) { delete (Indexer*)h; }

// m_unique_pos, m_dupe_pos (indexer.rs:71-72), number of keys, number of HIGH keys
void orc_index_stats(void* h, int64_t out[4]) {
  Indexer* ix = (Indexer*)h;
  out[0] = ix->m_unique_pos;
  out[1] = ix->m_dupe_pos;
  out[2] = (int64_t)ix->m_kmer_pos.size();
  int64_t high = 0;
  for (const auto& kv : ix->m_kmer_pos) high += kv.second.contig == DUPE_HIGH_LEVEL;
  out[3] = high;
}

// All keys of m_kmer_pos (unordered). `cap` entries available in `keys`.
int64_t orc_index_keys(void* h, int64_t* keys, int64_t cap) {
  Indexer* ix = (Indexer*)h;
  int64_t n = 0;
  for (const auto& kv : ix->m_kmer_pos) {
    if (n < cap) keys[n] = kv.first;
    ++n;
  }
  return n;
}

// Sites stored for one k-mer: returns 0 = absent (bitmap bit clear), -2 = HIGH,
// else the number of sites (1 or 2..5) written to contig[]/position[].
int32_t orc_index_lookup(void* h, int64_t kmer, int16_t contig[5], int32_t position[5]) {
  Indexer* ix = (Indexer*)h;
  if (kmer < 0 || kmer > 0xFFFFFFFFLL) return 0;
  if ((ix->m_bloom_filter[kmer >> 3] & (1u << (kmer & 7))) == 0) return 0;
  const GenePos& gp = ix->m_kmer_pos.at(kmer);
  if (gp.contig == DUPE_HIGH_LEVEL) return -2;
  if (gp.contig == DUPE_NORMAL_LEVEL) {
    const auto& lst = ix->m_dupe_list[gp.position];
    for (size_t g = 0; g < lst.size(); ++g) {
      contig[g] = lst[g].contig;
      position[g] = lst[g].position;
    }
    return (int32_t)lst.size();
  }
  contig[0] = gp.contig;
  position[0] = gp.position;
  return 1;
}

// m_fusion_seq[c] (indexer.rs:170); returns its length, copies up to cap bytes.
int64_t orc_index_fusion_seq(void* h, int32_t c, char* out, int64_t cap) {
  Indexer* ix = (Indexer*)h;
  const std::string& s = ix->m_fusion_seq.at((size_t)c);
  if (out) memcpy(out, s.data(), (size_t)std::min<int64_t>(cap, (int64_t)s.size()));
  return (int64_t)s.size();
}

static int32_t emit(const std::vector<SeqMatch>& v, orc_seqmatch* out) {
  for (size_t k = 0; k < v.size(); ++k) {
    out[k].seq_start = v[k].seq_start;
    out[k].seq_end = v[k].seq_end;
    out[k].position = v[k].start_gp.position;
    out[k].contig = v[k].start_gp.contig;
    out[k].pad = 0;
  }
  return (int32_t)v.size();
}

int32_t orc_map_read(void* h, const char* seq, int64_t len, orc_seqmatch out[2]) {
  return emit(((Indexer*)h)->map_read(seq, (int32_t)len), out);
}

// Batch form with the reference's threading shape: `threads` workers pull packs of
// 1000 reads (common.rs:23 PACK_SIZE) from a shared queue and call map_read once
// per read (pescanner.rs:296-311, :430).  threads <= 1 runs inline.
void orc_map_reads(void* h, const char* bases, const int64_t* offsets, int64_t n, int32_t threads,
                   int32_t* out_counts, orc_seqmatch* out_matches) {
  const Indexer* ix = (const Indexer*)h;
  const int64_t PACK = 1000;
  std::atomic<int64_t> next{0};
  auto worker = [&]() {
    for (;;) {
      int64_t p0 = next.fetch_add(PACK);
      if (p0 >= n) return;
      int64_t p1 = std::min(n, p0 + PACK);
      for (int64_t r = p0; r < p1; ++r) {
        auto v = ix->map_read(bases + offsets[r], (int32_t)(offsets[r + 1] - offsets[r]));
        out_counts[r] = emit(v, out_matches + 2 * r);
      }
    }
  };
  if (threads <= 1) {
    worker();
  } else {
    std::vector<std::thread> ts;
    for (int32_t t = 0; t < threads; ++t) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
  }
}

int64_t orc_make_kmer(const char* seq, int32_t pos, int64_t last_kmer, int32_t step) {
  return make_kmer(seq, pos, last_kmer, step);
}

int64_t orc_gp_to_i64(int16_t contig, int32_t position) { return gp_to_i64(GenePos{contig, position}); }

void orc_i64_to_gp(int64_t v, int16_t* contig, int32_t* position) {
  GenePos gp = i64_to_gp(v);
  *contig = gp.contig;
  *position = gp.position;
}

int32_t orc_segment_mask(const uint8_t* mask, int32_t seqlen, int16_t c1, int32_t p1, int16_t c2,
                         int32_t p2, orc_seqmatch out[2]) {
  return emit(segment_mask(mask, seqlen, GenePos{c1, p1}, GenePos{c2, p2}), out);
}

void orc_reverse_complement(const char* in, int64_t len, char* out) {
  std::string r = reverse_complement(std::string(in, (size_t)len));
  memcpy(out, r.data(), (size_t)len);
}

// src/core/indexer.rs:541-608.  `reversed[c]` = Fusion::is_reversed() of gene c.
int32_t orc_in_required_direction(const orc_seqmatch* m, int32_t n, const uint8_t* reversed) {
  if (n < 2) return 0;
  const orc_seqmatch* left = &m[0];
  const orc_seqmatch* right = &m[1];
  if (left->seq_start > right->seq_start) std::swap(left, right);
  if (left->position > 0 && right->position > 0) return 1;
  if (left->position < 0 && right->position < 0) return 0;
  bool lrev = reversed[left->contig] != 0, rrev = reversed[right->contig] != 0;
  if (lrev && !rrev) return 0;
  if (!lrev && rrev) return 1;
  if (left->contig < right->contig) return 1;
  // :597-599 tests `left.position.abs() < left.position.abs()` (left against
  // itself), which is always false, so the same-contig branch never returns true.
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// the immediate caller, FusionMapper::map_read + make_match +
// calc_distance + calc_ed (src/core/fusion_mapper.rs:93-251) and edit_distance
// (src/core/edit_distance.rs:12-197), restated literally.  Test infrastructure only.
// ---------------------------------------------------------------------------
namespace {

// src/core/edit_distance.rs:12-92 (Hyyro/Myers bit-parallel, N 64-bit blocks)
size_t edit_distance_bpv(std::map<char, std::vector<uint64_t>>& cmap, const std::string& v, size_t vsize,
                         size_t tmax, size_t tlen, size_t N) {
  size_t d = tmax * 64 + tlen;
  const uint64_t top = 1ull << ((tlen - 1) & 63);
  const uint64_t lmb = 1ull << 63;
  std::vector<uint64_t> d0(N, 0), hp(N, 0), hn(N, 0), vp(tmax + 1, 0), vn(tmax + 1, 0);
  for (size_t i = 0; i < tmax; ++i) vp[i] = ~0ull;
  for (size_t i = 0; i < tlen; ++i) vp[tmax] |= 1ull << (i & 63);
  for (size_t i = 0; i < vsize; ++i) {
    auto it = cmap.find(v[i]);
    if (it == cmap.end()) it = cmap.emplace(v[i], std::vector<uint64_t>(N, 0)).first;
    const std::vector<uint64_t>& pm = it->second;
    for (size_t r = 0; r <= tmax; ++r) {
      uint64_t x = pm[r];
      if (r > 0 && (hn[r - 1] & lmb) != 0) x |= 1ull;
      d0[r] = (((x & vp[r]) + vp[r]) ^ vp[r]) | x | vn[r];
      hp[r] = vn[r] | ~(d0[r] | vp[r]);
      hn[r] = d0[r] & vp[r];
      x = hp[r] << 1;
      if (r == 0 || (hp[r - 1] & lmb) != 0) x |= 1ull;
      vp[r] = (hn[r] << 1) | ~(d0[r] | x);
      if (r > 0 && (hn[r - 1] & lmb) != 0) vp[r] |= 1;
      vn[r] = d0[r] & x;
    }
    if ((hp[tmax] & top) != 0) d += 1;
    else if ((hn[tmax] & top) != 0) d -= 1;
  }
  return d;
}

// src/core/edit_distance.rs:94-120
size_t edit_distance_dp(const std::string& s1, size_t n1, const std::string& s2, size_t n2) {
  std::vector<std::vector<uint32_t>> d(n1 + 1, std::vector<uint32_t>(n2 + 1, 0));
  for (size_t i = 0; i <= n1; ++i) d[i][0] = (uint32_t)i;
  for (size_t j = 0; j <= n2; ++j) d[0][j] = (uint32_t)j;
  for (size_t i = 1; i <= n1; ++i)
    for (size_t j = 1; j <= n2; ++j)
      d[i][j] = std::min(std::min(d[i - 1][j], d[i][j - 1]) + 1, d[i - 1][j - 1] + (s1[i - 1] == s2[j - 1] ? 0u : 1u));
  return d[n1][n2];
}

// src/core/edit_distance.rs:122-157
size_t edit_distance_map(const std::string& a, size_t asize, const std::string& b, size_t bsize, size_t N) {
  std::map<char, std::vector<uint64_t>> cmap;
  const size_t tmax = (asize - 1) >> 6;
  const size_t tlen = asize - tmax * 64;
  for (size_t i = 0; i < tmax; ++i)
    for (size_t j = 0; j < 64; ++j) {
      auto it = cmap.emplace(a[i * 64 + j], std::vector<uint64_t>(N, 0)).first;
      it->second[i] |= 1ull << j;
    }
  for (size_t i = 0; i < tlen; ++i) {
    auto it = cmap.emplace(a[tmax * 64 + i], std::vector<uint64_t>(N, 0)).first;
    it->second[tmax] |= 1ull << i;
  }
  return edit_distance_bpv(cmap, b, bsize, tmax, tlen, N);
}

// src/core/edit_distance.rs:159-193
size_t edit_distance(std::string a, size_t asize, std::string b, size_t bsize) {
  if (asize == 0) return bsize;
  if (bsize == 0) return asize;
  if (asize < bsize) { std::swap(a, b); std::swap(asize, bsize); }
  size_t vsize = ((asize - 1) >> 6) + 1;
  if (vsize > 10) {
    std::swap(a, b);
    std::swap(asize, bsize);
    vsize = ((asize - 1) >> 6) + 1;
  }
  if (vsize >= 1 && vsize <= 10) return edit_distance_map(a, asize, b, bsize, vsize);
  return edit_distance_dp(a, asize, b, bsize);
}

// src/core/fusion_mapper.rs:225-251
int32_t calc_ed(const std::vector<std::string>& fusion_seq, const std::string& seq, int32_t contig, int32_t start,
                int32_t end) {
  if ((start >= 0 && end <= 0) || (start <= 0 && end >= 0)) return -1;
  const std::string& fs = fusion_seq.at((size_t)contig);
  if (std::abs(start) >= (int32_t)fs.size() || std::abs(end) >= (int32_t)fs.size()) return -2;
  std::string ss = seq;
  if (start < 0) {
    ss = reverse_complement(seq);
    int32_t tmp = start;
    start = -end;
    end = -tmp;
  }
  std::string ref_str = fs.substr((size_t)start, (size_t)(end - start + 1));
  return (int32_t)edit_distance(ss, ss.size(), ref_str, ref_str.size());
}

}  // namespace

extern "C" {

struct orc_readmatch {
  int32_t read_break, gap, left_distance, right_distance;
  int32_t left_position, right_position;
  int16_t left_contig, right_contig;
};

int64_t orc_edit_distance(const char* a, int64_t alen, const char* b, int64_t blen) {
  return (int64_t)edit_distance(std::string(a, (size_t)alen), (size_t)alen, std::string(b, (size_t)blen), (size_t)blen);
}

// FusionMapper::map_read after m_indexer.map_read (fusion_mapper.rs:100-131):
// returns 0 = None with mapable=false, 1 = None with mapable=true, 2 = Some(ReadMatch).
int32_t orc_fusion_map_read(void* h, const uint8_t* reversed, const char* seq, int64_t len,
                            const orc_seqmatch* mapping, int32_t n_mapping, orc_readmatch* out) {
  Indexer* ix = (Indexer*)h;
  if (n_mapping < 2) return 0;                                         // :107-115
  if (!orc_in_required_direction(mapping, n_mapping, reversed)) return 1;  // :118-123
  if (n_mapping != 2) return 1;                                        // make_match :155-157 (None, mapable stays true)
  orc_seqmatch left = mapping[0], right = mapping[1];                  // :160-166
  if (left.seq_start > right.seq_start) std::swap(left, right);
  const int32_t read_break = (left.seq_end + right.seq_start) / 2;     // :173
  left.position += read_break;                                         // :177-178
  right.position += read_break + 1;
  const int32_t gap = right.seq_start - left.seq_end - 1;              // :180
  const std::string s(seq, (size_t)len);
  const int32_t left_len = read_break + 1;                             // calc_distance :199-221
  const int32_t right_len = (int32_t)len - (read_break + 1);
  const std::string left_seq = s.substr(0, (size_t)left_len);
  const std::string right_seq = s.substr((size_t)(read_break + 1), (size_t)right_len);
  out->read_break = read_break;
  out->gap = gap;
  out->left_contig = left.contig;
  out->left_position = left.position;
  out->right_contig = right.contig;
  out->right_position = right.position;
  out->left_distance = calc_ed(ix->m_fusion_seq, left_seq, left.contig, left.position - left_len + 1, left.position);
  out->right_distance = calc_ed(ix->m_fusion_seq, right_seq, right.contig, right.position, right.position + right_len - 1);
  return 2;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// SequenceReadPair::fast_merge (src/core/read.rs:313-440), restated
// literally.  Pinned by the reference's own asserting test (read.rs:450-486).
// ---------------------------------------------------------------------------
extern "C" {

// left/right as read from the FASTQ pair (right NOT yet reverse-complemented).
// Returns 1 and fills out_seq/out_qual (capacity len1 + len2), *out_len, *out_diff when the
// pair merges, 0 otherwise.
int32_t orc_fast_merge(const char* l_seq, const char* l_qual, int32_t len1, const char* r_seq, const char* r_qual,
                       int32_t len2, char* out_seq, char* out_qual, int32_t* out_len, int32_t* out_diff) {
  // :314 rc_right = m_right.reverse_complement(): sequence reverse-complemented, quality reversed
  const std::string str1(l_seq, (size_t)len1), qual1(l_qual, (size_t)len1);
  const std::string str2 = reverse_complement(std::string(r_seq, (size_t)len2));
  std::string qual2(r_qual, (size_t)len2);
  for (int32_t i = 0; i < len2 / 2; ++i) std::swap(qual2[(size_t)i], qual2[(size_t)(len2 - 1 - i)]);
  const int32_t MIN_OVERLAP = 30;  // :325
  bool overlapped = false;
  int32_t olen = MIN_OVERLAP, diff = 0, low_qual_diff = 0;
  while (olen <= std::min(len1, len2)) {  // :339-367
    diff = 0;
    low_qual_diff = 0;
    bool ok = true;
    const int32_t offset = len1 - olen;
    for (int32_t i = 0; i < olen; ++i) {
      if (str1[(size_t)(offset + i)] != str2[(size_t)i]) {
        diff += 1;
        if ((qual1[(size_t)(offset + i)] >= '?' && qual2[(size_t)i] <= '0') ||
            (qual1[(size_t)(offset + i)] <= '0' && qual2[(size_t)i] >= '?'))
          low_qual_diff += 1;
        if (diff > low_qual_diff || low_qual_diff >= 3) {
          ok = false;
          break;
        }
      }
    }
    if (ok) {
      overlapped = true;
      break;
    }
    olen += 1;
  }
  if (!overlapped) return 0;
  const int32_t offset = len1 - olen;  // :370
  std::string ms = str1.substr(0, (size_t)offset) + str2;   // :380-386
  std::string mq = qual1.substr(0, (size_t)offset) + qual2;  // :393-399
  for (int32_t i = 0; i < olen; ++i) {  // :402-428
    if (str1[(size_t)(offset + i)] != str2[(size_t)i]) {
      if (qual1[(size_t)(offset + i)] >= '?' && qual2[(size_t)i] <= '0') {
        ms[(size_t)(offset + i)] = str1[(size_t)(offset + i)];
        mq[(size_t)(offset + i)] = qual1[(size_t)(offset + i)];
      } else {
        ms[(size_t)(offset + i)] = str2[(size_t)i];
        mq[(size_t)(offset + i)] = qual2[(size_t)i];
      }
    } else {
      uint32_t q = (uint32_t)(unsigned char)qual1[(size_t)(offset + i)] + (uint32_t)(unsigned char)qual2[(size_t)i] - 33u;
      mq[(size_t)(offset + i)] = q >= (uint32_t)'Z' ? 'Z' : (char)q;
    }
  }
  memcpy(out_seq, ms.data(), ms.size());
  memcpy(out_qual, mq.data(), mq.size());
  *out_len = (int32_t)ms.size();
  *out_diff = diff;
  return 1;
}

// ---- FastqReader::read (src/core/fastq_reader.rs:75-147) over an
// in-memory text, one read_line at a time like the reference.  out[8*i ..] = (start, length)
// of the name, sequence, strand and quality lines of record i; returns the number of
// records (writes at most cap of them).
static bool orc_read_line(const char* text, int64_t n, int64_t& cur, int64_t& start, int64_t& len) {
  if (cur >= n) return false;  // read_line returned 0 bytes (:80, :94, :108, :123)
  start = cur;
  int64_t p = cur;
  while (p < n && text[p] != '\n') ++p;
  len = p - cur;                     // the trailing newline is popped if it is there (:82-87)
  cur = p < n ? p + 1 : p;
  return true;
}

int64_t orc_fastq_cut(const char* text, int64_t n, int64_t* out, int64_t cap) {
  int64_t cur = 0, n_rec = 0;
  for (;;) {
    int64_t st[4], ln[4];
    bool ok = true;
    for (int k = 0; k < 4 && ok; ++k) ok = orc_read_line(text, n, cur, st[k], ln[k]);
    if (!ok) break;  // read() returns None: the scanners stop
    if (n_rec < cap)
      for (int k = 0; k < 4; ++k) {
        out[8 * n_rec + 2 * k] = st[k];
        out[8 * n_rec + 2 * k + 1] = ln[k];
      }
    ++n_rec;
  }
  return n_rec;
}

}  // extern "C"
