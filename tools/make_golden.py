"""Generate tests/golden/branch_cases.json: the branch-coverage inputs of
tests/helpers.py with the outputs of the C++ oracle, after checking that the
independent Python model (oracle/indexer_model.py) returns exactly the same.

The reference (Rust) cannot be run in this pipeline, so these vectors are NOT
reference outputs: they are "two independent restatements agree" vectors
(parity unpinned).  Inputs and expected outputs only.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import indexer_model as M  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import branch_genes, branch_reads  # noqa: E402


def main():
    genes, rev = branch_genes()
    reads = branch_reads(genes)
    ox = O.OracleIndexer(genes)
    mx = M.IndexModel([None if g is None else g.decode() for g in genes])
    cases = []
    n_nonempty = 0
    for label, r in reads:
        a = ox.map_read(r)
        b = mx.map_read(r.decode())
        assert a == b, (label, a, b)
        n_nonempty += bool(a)
        cases.append({"label": label, "read": r.decode(), "expect": a})
    # index content: every key of the oracle, with its sites
    keys = sorted(int(k) for k in ox.keys())
    assert set(keys) == set(mx.table.keys())
    index = []
    for k in keys:
        n, sites = ox.lookup(k)
        v = mx.table[k]
        if n == -2:
            assert v is M.HIGH
        else:
            assert sorted(v) == sites, (k, v, sites)
        index.append([k, n, sites])
    out = {
        "note": "outputs of oracle/indexer_oracle.cc, equal to oracle/indexer_model.py; not reference outputs",
        "genes": [None if g is None else g.decode() for g in genes],
        "reversed": rev,
        "stats": ox.stats(),
        "index": index,
        "cases": cases,
    }
    path = os.path.join(ROOT, "tests", "golden", "branch_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, len(cases), "cases,", n_nonempty, "non-empty,", len(keys), "keys,",
          os.path.getsize(path), "bytes")
    labels = [c["label"] for c in cases if c["expect"] and not c["label"].startswith("random")]
    print("non-empty named:", labels)
    for c in cases:
        if not c["label"].startswith(("random", "background")):
            print("%-32s %s" % (c["label"], c["expect"]))


if __name__ == "__main__":
    main()
