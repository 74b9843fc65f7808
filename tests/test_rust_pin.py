"""The maintainer-side pin (tests/rust/indexer_golden.rs + tests/golden/branch_cases.tsv) stays in step with the JSON
fixtures, and the text form says what the JSON says: it is parsed here with the grammar the Rust module uses and
checked against the oracle the way the Rust module checks the reference's Indexer."""
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_tsv_and_label_list_are_regenerated_from_the_json():
    from tools import make_rust_pin as P
    assert open(os.path.join(G, "branch_cases.tsv")).read() == P.tsv_text(), "run tools/make_rust_pin.py"
    rs = open(os.path.join(ROOT, "tests", "rust", "indexer_golden.rs")).read()
    assert P.labels_block() in rs, "run tools/make_rust_pin.py"
    labels = [c["label"] for c in json.load(open(os.path.join(G, "branch_cases.json")))["cases"]]
    named = re.findall(r'"([A-Za-z0-9_]+)",', rs[rs.index("const LABELS"):rs.index("// END GENERATED")])
    assert named == labels and len(labels) == 246


def test_rust_module_tests_what_it_claims():
    rs = open(os.path.join(ROOT, "tests", "rust", "indexer_golden.rs")).read()
    for needle in ("fn golden_make_index_statistics_and_every_key", "fn golden_map_read_all_branch_cases",
                   "fn golden_fast_merge_and_edit_distance", "ix.make_index()", "ix.map_read(&r)", ".fast_merge()",
                   "m_unique_pos", "m_dupe_pos", "m_dupe_list", "m_bloom_filter", "edit_distance_from_str"):
        assert needle in rs, needle
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "tests/rust/indexer_golden.rs" in doc and "branch_cases.tsv" in doc and "cargo test golden" in doc


def _field(s):
    return "" if s == "-" else s


def test_text_form_against_the_oracle_like_the_rust_module_does(oracle):
    genes, stats, keys, cases, merge, edit = [], {}, [], [], [], []
    for line in open(os.path.join(G, "branch_cases.tsv")).read().splitlines():
        if not line or line.startswith("#"):
            continue
        f = line.split("\t")
        if f[0] == "GENE":
            genes.append(None if f[3] == "-" else f[3].encode())
        elif f[0] == "STAT":
            stats[f[1]] = int(f[2])
        elif f[0] == "KEY":
            keys.append((int(f[1]), int(f[2]), [tuple(int(x) for x in s.split(":")) for s in _field(f[3]).split(";") if s]))
        elif f[0] == "CASE":
            exp = [tuple(int(x) for x in s.split(":")) for s in _field(f[4]).split(";") if s]
            assert len(exp) == int(f[3])
            cases.append((f[1], _field(f[2]).encode(), exp))
        elif f[0] == "MERGE":
            merge.append(f[1:6])
        elif f[0] == "EDIT":
            edit.append((_field(f[1]), _field(f[2]), int(f[3])))
        else:
            raise AssertionError(f[0])
    ox = oracle.OracleIndexer(genes)
    assert ox.stats() == stats
    assert sorted(int(k) for k in ox.keys()) == [k for k, _, _ in keys]
    for k, n, sites in keys[::7]:
        gn, gs = ox.lookup(k)
        assert gn == n and [tuple(s) for s in gs] == sites
    for label, read, exp in cases:
        assert ox.map_read(read) == exp, label
    for ls, lq, rs_, rq, ms in merge:
        got = oracle.fast_merge(ls.encode(), lq.encode(), rs_.encode(), rq.encode())
        assert got is not None and got[0].decode() == ms
    for a, b, d in edit:
        assert oracle.edit_distance(a.encode(), b.encode()) == d
    assert len(cases) == 246 and len(merge) == 1 and len(edit) == 3
