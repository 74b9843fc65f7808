"""Derive the synthetic index shapes (gene names + spans) from the reference's
own test CSVs.  Run once in the build container (needs /root/reference); the
output JSON is committed because /root/reference does not exist on the GPU box.

Shapes:
  IDX-T  testdata/fusions.csv, 4 genes
  IDX-D  first 32 genes of testdata/cancer.csv ("druggable.hg38-shaped"; the
         upstream druggable.hg38.csv is git-ignored and unavailable offline)
  IDX-C  all 136 genes of testdata/cancer.csv ("cancer.hg38-shaped")
Only gene name, chromosome, start, end and the exon-order flag are kept.
"""
import json
import sys


def parse(path):
    genes = []
    cur = None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                name, loc = line[1:].split(",")
                chrom, span = loc.split(":")
                start, end = span.split("-")
                cur = {"name": name, "chr": chrom, "start": int(start), "end": int(end), "exons": []}
                genes.append(cur)
            else:
                eid, s, e = line.split(",")
                cur["exons"].append((int(s), int(e)))
    out = []
    for g in genes:
        ex = g["exons"]
        # gene.rs:98-107: reversed when exon 1 starts after exon 2
        rev = len(ex) >= 2 and ex[0][0] > ex[1][0]
        out.append({"name": g["name"], "chr": g["chr"], "start": g["start"], "end": g["end"],
                    "len": g["end"] - g["start"], "reversed": bool(rev)})
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    fus = parse(ref + "/testdata/fusions.csv")
    can = parse(ref + "/testdata/cancer.csv")
    shapes = {"IDX-T": fus, "IDX-D": can[:32], "IDX-C": can}
    for k, v in shapes.items():
        print(k, len(v), "genes", sum(g["len"] for g in v), "bp", file=sys.stderr)
    json.dump(shapes, open("genefuserust_amd/data/index_shapes.json", "w"), indent=0)
