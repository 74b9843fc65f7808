// C++ host-side mirror of the reference's `Indexer` (src/core/indexer.rs:67-608)
// over the gfmatch C ABI.  Header-only; link with libgfmatch.so.
//
// The reference is Rust and this image has no Rust toolchain, so this is the
// compiled-language host side above the C ABI: same names, argument meaning and
// error behaviour as the Rust type (construction errors are returned as
// exceptions where Rust returns Result; hot calls that "cannot fail" in Rust
// throw only when the device is lost).  INTEGRATION.md shows the equivalent
// Rust binding.
#pragma once

#include <cstdint>
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gfmatch.h"

namespace genefuse {

struct GenePos {  // src/core/common.rs:4-7
  int16_t contig = 0;
  int32_t position = 0;
};

struct SeqMatch {  // src/core/indexer.rs:41-45
  int32_t seq_start = 0;
  int32_t seq_end = 0;
  GenePos start_gp;
};

struct Exon {  // src/core/gene.rs:9-13
  int32_t id = 0, start = 0, end = 0;
};

struct Gene {  // src/core/gene.rs:16-229
  std::string m_name = "invalid", m_chr = "invalid";
  int32_t m_start = 0, m_end = 0;
  bool m_reversed = false;
  std::vector<Exon> m_exons;

  Gene() = default;
  Gene(std::string name, std::string chr, int32_t start, int32_t end, bool reversed = false)
      : m_name(std::move(name)), m_chr(std::move(chr)), m_start(start), m_end(end), m_reversed(reversed) {}

  bool is_reversed() const { return m_reversed; }
  bool valid() const { return m_name != "invalid" && m_start != 0 && m_end != 0; }  // :39-41

  void add_exon(int32_t id, int32_t start, int32_t end) {  // :90-105
    m_exons.push_back(Exon{id, start, end});
    if (m_exons.size() > 1 && m_exons[0].start > m_exons[1].start) m_reversed = true;
  }

  // :131-169, e.g. "ALK:exon:20|-chr2:29446222"
  std::string pos2str(int32_t pos) const {
    const int32_t pp = (pos < 0 ? -pos : pos) + m_start;
    std::string ss = m_name + ":";
    for (size_t i = 0; i < m_exons.size(); ++i) {
      const Exon& e = m_exons[i];
      if (pp >= e.start && pp <= e.end) {
        ss += "exon:" + std::to_string(e.id) + "|";
        break;
      }
      if (i > 0) {
        const Exon& p = m_exons[i - 1];
        if (m_reversed ? (e.end < pp && pp < p.start) : (p.end < pp && pp < e.start)) {
          ss += "intron:" + std::to_string(e.id - 1) + "|";
          break;
        }
      }
    }
    ss += pos >= 0 ? "+" : "-";
    return ss + m_chr + ":" + std::to_string(pp);
  }

  // :171-203; the out-parameters stay as they were when no exon or intron holds the position.
  // (The reference never advances its prev_exon: introns are measured against the FIRST exon.)
  void get_exon_intron(int32_t pos, bool& is_exon, int32_t& number) const {
    const int32_t pp = (pos < 0 ? -pos : pos) + m_start;
    for (size_t i = 0; i < m_exons.size(); ++i) {
      const Exon& e = m_exons[i];
      if (pp >= e.start && pp <= e.end) {
        is_exon = true;
        number = e.id;
        return;
      }
      if (i > 0) {
        const Exon& p = m_exons[0];
        if (m_reversed ? (e.end < pp && pp < p.start) : (p.end < pp && pp < e.start)) {
          is_exon = false;
          number = e.id - 1;
          return;
        }
      }
    }
  }

  int32_t gene_pos_2_chr_pos(int32_t genepos) const {  // :205-212
    const int32_t chrpos = (genepos < 0 ? -genepos : genepos) + m_start;
    return genepos < 0 ? -chrpos : chrpos;
  }
};

namespace detail {
inline std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
inline std::vector<std::string> split(const std::string& s, char d) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    const size_t b = s.find(d, a);
    out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
    if (b == std::string::npos) break;
    a = b + 1;
  }
  return out;
}
// Rust's str::parse::<i32> on a trimmed field: optional sign, digits, 32-bit range; throws otherwise
inline int32_t parse_i32(const std::string& field) {
  const std::string t = trim(field);
  size_t k = (!t.empty() && (t[0] == '+' || t[0] == '-')) ? 1 : 0;
  if (k == t.size()) throw std::runtime_error("invalid digit found in string: '" + t + "'");
  long long v = 0;
  for (size_t i = k; i < t.size(); ++i) {
    if (t[i] < '0' || t[i] > '9') throw std::runtime_error("invalid digit found in string: '" + t + "'");
    v = v * 10 + (t[i] - '0');
    if (v > 2147483648LL) throw std::runtime_error("number too large to fit in target type: '" + t + "'");
  }
  if (t[0] == '-') v = -v;
  if (v > 2147483647LL) throw std::runtime_error("number too large to fit in target type: '" + t + "'");
  return (int32_t)v;
}
}  // namespace detail

inline Gene gene_parse(const std::string& line_str) {  // Gene::parse, gene.rs:43-88
  const auto splitted = detail::split(line_str, ',');
  if (splitted.size() < 2) return Gene();
  const std::string name = detail::trim(splitted[0].substr(splitted[0].empty() ? 0 : 1));
  const auto chr_pos = detail::split(splitted[1], ':');
  if (chr_pos.size() < 2) return Gene();
  const auto range = detail::split(chr_pos[1], '-');
  if (range.size() < 2) return Gene();
  return Gene(name, detail::trim(chr_pos[0]), detail::parse_i32(range[0]), detail::parse_i32(range[1]));
}

struct Fusion {  // src/core/fusion.rs:12-107
  Gene m_gene;
  bool is_reversed() const { return m_gene.is_reversed(); }
  std::string pos2str(int32_t pos) const { return m_gene.pos2str(pos); }

  // Fusion::parse_csv (:22-86) on the file's text
  static std::vector<Fusion> parse_csv_text(const std::string& text) {
    std::vector<Fusion> fusions;
    Gene working;
    for (const std::string& raw : detail::split(text, '\n')) {
      const std::string line = detail::trim(raw);
      const auto splitted = detail::split(line, ',');
      if (splitted.size() < 2 || (!splitted[0].empty() && splitted[0][0] == '#')) continue;
      if (!splitted[0].empty() && splitted[0][0] == '>') {
        if (working.valid()) fusions.push_back(Fusion{working});
        working = gene_parse(line);
        continue;
      }
      if (splitted.size() < 3) continue;
      working.add_exon(detail::parse_i32(splitted[0]), detail::parse_i32(splitted[1]), detail::parse_i32(splitted[2]));
    }
    if (working.valid()) fusions.push_back(Fusion{working});
    return fusions;
  }
  static std::vector<Fusion> parse_csv(const std::string& filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_csv_text(ss.str());
  }
};

// FastaReader::read_all (fasta_reader.rs:117-200) on the (already gunzipped) bytes of a file:
// name -> sequence.  A record runs from one '>' to the next; its name ends at the first
// newline or blank; of the rest only letters, '-' and '*' are kept, upper-cased on request.
inline std::map<std::string, std::string> fasta_read_all(const std::string& data, bool force_upper_case = true) {
  std::map<std::string, std::string> contigs;
  size_t pos = data.find('>');
  if (pos == std::string::npos) return contigs;
  ++pos;
  while (pos < data.size()) {
    const size_t next = data.find('>', pos);
    const std::string rec = data.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
    size_t k = 0;
    while (k < rec.size() && rec[k] != '\n' && rec[k] != ' ') ++k;
    std::string seq;
    for (size_t i = k + 1; i < rec.size(); ++i) {
      const unsigned char b = (unsigned char)rec[i];
      if ((b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '-' || b == '*')
        seq.push_back(force_upper_case && b >= 'a' && b <= 'z' ? (char)(b - 32) : (char)b);
    }
    contigs[rec.substr(0, k)] = seq;
    if (next == std::string::npos) break;
    pos = next + 1;
  }
  return contigs;
}

class Indexer {
 public:
  using Contigs = std::map<std::string, std::string>;  // FastaReader.m_all_contigs (fasta_reader.rs:35)

  // Indexer::with_loaded_ref (indexer.rs:100-112)
  static Indexer with_loaded_ref(const Contigs* reference, std::vector<Fusion> fusions, int device = -1) {
    Indexer ix;
    ix.m_reference = reference;
    ix.m_fusions = std::move(fusions);
    ix.device_ = device;
    return ix;
  }

  Indexer() = default;
  Indexer(const Indexer&) = delete;
  Indexer& operator=(const Indexer&) = delete;
  Indexer(Indexer&& o) noexcept { *this = std::move(o); }
  Indexer& operator=(Indexer&& o) noexcept {
    if (this != &o) {
      release();
      h_ = o.h_;
      o.h_ = nullptr;
      m_reference = o.m_reference;
      m_fusions = std::move(o.m_fusions);
      m_fusion_seq = std::move(o.m_fusion_seq);
      device_ = o.device_;
    }
    return *this;
  }
  ~Indexer() { release(); }

  const Contigs* get_ref() const { return m_reference; }  // indexer.rs:114

  // Indexer::make_index (indexer.rs:122-177)
  void make_index() {
    if (!m_reference) return;  // :123-125
    std::vector<const char*> ptr(m_fusions.size(), nullptr);
    std::vector<int64_t> len(m_fusions.size(), -1);
    for (size_t c = 0; c < m_fusions.size(); ++c) {
      const Gene& g = m_fusions[c].m_gene;
      std::string chr = g.m_chr;  // :137-152
      auto it = m_reference->find(chr);
      if (it == m_reference->end()) it = m_reference->find("chr" + chr);
      if (it == m_reference->end()) {
        std::string stripped = chr;
        for (size_t p; (p = stripped.find("chr")) != std::string::npos;) stripped.erase(p, 3);
        it = m_reference->find(stripped);
      }
      if (it == m_reference->end()) continue;  // gene indexes nothing, m_fusion_seq[c] = ""
      const std::string& s = it->second;
      if (g.m_start < 0 || g.m_end < g.m_start || (size_t)g.m_end > s.size())
        throw std::out_of_range("gene " + g.m_name + ": slice outside its contig");  // .get(a..b).unwrap(), :157-158
      ptr[c] = s.data() + g.m_start;
      len[c] = g.m_end - g.m_start;
    }
    gf_options opts{};
    opts.device = device_;
    gf_index* h = nullptr;
    check(gf_index_build(ptr.data(), len.data(), (int32_t)m_fusions.size(), &opts, &h));
    release();
    h_ = h;
    m_fusion_seq.assign(m_fusions.size(), std::string());
    for (size_t c = 0; c < m_fusions.size(); ++c) {
      int64_t n = gf_index_fusion_seq(h_, (int32_t)c, nullptr, 0);
      m_fusion_seq[c].resize((size_t)n);
      if (n > 0) gf_index_fusion_seq(h_, (int32_t)c, &m_fusion_seq[c][0], n);
    }
  }

  // Indexer::map_read (indexer.rs:252-538): one read per call
  std::vector<SeqMatch> map_read(const std::string& seq) const {
    gf_seqmatch out[2];
    int n = check(gf_map_read(handle(), seq.data(), (int64_t)seq.size(), out));
    std::vector<SeqMatch> v;
    for (int k = 0; k < n; ++k) v.push_back(from_c(out[k]));
    return v;
  }

  // the batch form the GPU wants: one call per pack set (pescanner.rs:430-515 restructured)
  std::vector<std::vector<SeqMatch>> map_reads(const std::vector<std::string>& reads) const {
    std::string bases;
    std::vector<int64_t> off(reads.size() + 1, 0);
    for (size_t r = 0; r < reads.size(); ++r) {
      bases += reads[r];
      off[r + 1] = (int64_t)bases.size();
    }
    std::vector<int32_t> counts(reads.size());
    std::vector<gf_seqmatch> m(2 * reads.size());
    check(gf_map_reads(handle(), bases.data(), off.data(), (int64_t)reads.size(), counts.data(), m.data()));
    std::vector<std::vector<SeqMatch>> out(reads.size());
    for (size_t r = 0; r < reads.size(); ++r)
      for (int k = 0; k < counts[r]; ++k) out[r].push_back(from_c(m[2 * r + k]));
    return out;
  }

  // Indexer::in_required_direction (indexer.rs:541-608)
  bool in_required_direction(const std::vector<SeqMatch>& mapping) const {
    std::vector<gf_seqmatch> m(mapping.size() ? mapping.size() : 1);
    for (size_t k = 0; k < mapping.size(); ++k)
      m[k] = gf_seqmatch{mapping[k].seq_start, mapping[k].seq_end, mapping[k].start_gp.position,
                         mapping[k].start_gp.contig, 0};
    std::vector<uint8_t> rev(m_fusions.size() ? m_fusions.size() : 1, 0);
    for (size_t c = 0; c < m_fusions.size(); ++c) rev[c] = m_fusions[c].is_reversed();
    return check(gf_in_required_direction(m.data(), (int32_t)mapping.size(), rev.data(), (int32_t)m_fusions.size())) != 0;
  }

  gf_index* handle() const {
    if (!h_) throw std::logic_error("make_index() has not been called");
    return h_;
  }

  const Contigs* m_reference = nullptr;   // Indexer.m_reference (indexer.rs:69)
  std::vector<Fusion> m_fusions;          // :70
  std::vector<std::string> m_fusion_seq;  // :77

 private:
  static SeqMatch from_c(const gf_seqmatch& c) {
    SeqMatch m;
    m.seq_start = c.seq_start;
    m.seq_end = c.seq_end;
    m.start_gp.contig = c.contig;
    m.start_gp.position = c.position;
    return m;
  }
  static int check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("gfmatch: ") + gf_last_error());
    return rc;
  }
  void release() {
    if (h_) gf_index_free(h_);
    h_ = nullptr;
  }
  gf_index* h_ = nullptr;
  int device_ = -1;
};

}  // namespace genefuse
