// C++ host test of include/gf_indexer.hpp, no GPU: Fusion::parse_csv / Gene::pos2str and the
// FASTA reader against the reference's own test vectors (fusion.rs:112-150 on
// testdata/fusions.csv, fasta_reader.rs:233-258 on testdata/tinyref.fa).
//   test_inputs <fusions.csv> <tinyref.fa>
#include <cstdio>

#include "gf_indexer.hpp"

using namespace genefuse;

#define EXPECT(cond)                                             \
  do {                                                           \
    if (!(cond)) {                                               \
      printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);     \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::vector<Fusion> fusions = Fusion::parse_csv(argv[1]);
  EXPECT(fusions.size() == 4);
  EXPECT(fusions[0].m_gene.m_name == "ALK" && fusions[3].m_gene.m_name == "EML4");
  EXPECT(fusions[0].is_reversed() && !fusions[3].is_reversed());
  EXPECT(fusions[0].pos2str(-30582) == "ALK:exon:20|-chr2:29446222");
  EXPECT(fusions[0].pos2str(31060) == "ALK:intron:19|+chr2:29446700");
  EXPECT(fusions[3].pos2str(95365) == "EML4:exon:6|+chr2:42491855");
  EXPECT(fusions[3].pos2str(95346) == "EML4:intron:5|+chr2:42491836");
  EXPECT(fusions[0].m_gene.gene_pos_2_chr_pos(-5) == -(5 + 29415640));
  std::ifstream f(argv[2], std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  const auto contigs = fasta_read_all(ss.str(), true);
  EXPECT(contigs.size() == 2);
  EXPECT(contigs.at("contig1") == "GATCACAGGTCTATCACCCTATTAATTGGTATTTTCGTCTGGGGGGTGTGGAGCCGGAGCACCCTATGTCGCAGT");
  EXPECT(contigs.at("contig2") == "GTCTGCACAGCCGCTTTCCACACAGAACCCCCCCCTCCCCCCGCTTCTGGCAAACCCCAAAAACAAAGAACCCTA");
  const auto odd = fasta_read_all("junk>a desc\nAC gt\n>b\n>", false);
  EXPECT(odd.size() == 2 && odd.at("a") == "descACgt" && odd.at("b").empty());
  bool threw = false;
  try {
    Fusion::parse_csv_text(">G,chr1:10-20\n1,,\n");
  } catch (const std::exception&) {
    threw = true;
  }
  EXPECT(threw);
  printf("ok: %zu genes, %zu contigs\n", fusions.size(), contigs.size());
  return 0;
}
