// Harness for include/gf_fusion_result.hpp: reads a fusion CSV, the gene sequences (one per
// line) and a match list (tab-separated, one match per line: name read qual break lc lp rc rp
// gap ld rd reversed), clusters, and prints report_text, a separator line, report_json.
// tests/test_fusion_result.py compares the output with the Python mirror's, byte for byte.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "gf_fusion_result.hpp"

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  using namespace genefuse;
  std::vector<Fusion> fusions = Fusion::parse_csv(argv[1]);
  std::vector<std::string> seqs;
  {
    std::ifstream in(argv[2]);
    std::string line;
    while (std::getline(in, line)) seqs.push_back(line);
  }
  std::vector<ReadMatch> ms;
  {
    std::ifstream in(argv[3]);
    std::string line;
    while (std::getline(in, line)) {
      std::vector<std::string> f = detail::split(line, '\t');
      if (f.size() < 12) return 3;
      ReadMatch m;
      m.m_name = f[0];
      m.m_read = f[1];
      m.m_quality = f[2];
      m.m_read_break = std::stoi(f[3]);
      m.m_left_gp = GenePos{(int16_t)std::stoi(f[4]), std::stoi(f[5])};
      m.m_right_gp = GenePos{(int16_t)std::stoi(f[6]), std::stoi(f[7])};
      m.m_gap = std::stoi(f[8]);
      m.m_left_distance = std::stoi(f[9]);
      m.m_right_distance = std::stoi(f[10]);
      m.m_reversed = f[11] == "1";
      ms.push_back(m);
    }
  }
  Settings st;
  st.unique_requirement = std::atoi(argv[4]);
  st.output_deletions = std::atoi(argv[5]) != 0;
  st.output_untranslated = std::atoi(argv[6]) != 0;
  std::vector<FusionResult> res = cluster_matches(group_and_sort(ms, fusions.size()), fusions, seqs, st);
  std::cout << report_text(res) << "\n====\n" << report_json(res, "cmd", "v", "t", st);
  return 0;
}
