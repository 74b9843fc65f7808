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
#include <map>
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

struct Gene {  // fields of src/core/gene.rs:16-23 read by the Indexer
  std::string m_name, m_chr;
  int32_t m_start = 0, m_end = 0;
  bool m_reversed = false;
  bool is_reversed() const { return m_reversed; }
};

struct Fusion {  // src/core/fusion.rs:14-16
  Gene m_gene;
  bool is_reversed() const { return m_gene.is_reversed(); }
};

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
