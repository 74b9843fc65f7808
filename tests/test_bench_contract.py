"""The bench line's contract fields, checked on the CPU against the committed r04 lines (profiles/r04_bench_config*.json are
what bench.py printed on the MI355X): metric / unit / scaling / roofline / cpu_baseline shape, that `roofline.frac` can be
recomputed from profiles/hbm_traffic.json and the line's own kernel time, and — VERDICT r03 — that the
counter entries the lines rest on were taken from THIS tree's kernels: every entry carries the commit, the hash of the
kernel sources and the hash of the library it was profiled with; a line made from other sources says `traffic_stale`,
and a tree whose sources have moved on since the profile fails here until it is profiled again."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(c):
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_config%d.json" % c)))


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_line_has_the_contract_fields(c):
    j = _line(c)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "reads/s" and j["higher_is_better"] is True and j["vs_baseline"] is None and j["dtype"] == "u32"
    assert "workload" in j["config"] and "BASELINE configs[%d]" % c in j["config"]["workload"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 0.9
    assert j["parity"] and all(v["bit_exact"] for v in (j["parity"].values() if c == 4 else [j["parity"]]))


def test_config1_line_carries_cpu_baseline_and_extras():
    j = _line(1)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "reads/s" and cb["value"] > 1e5 and cb["sample"]
    assert "pairs" in j["config"]["workload"] and "N(300,30)" in j["config"]["workload"]
    assert j["h2d_inclusive"]["packed"]["same_hits_as_ascii"] and j["packed_input"]["identical_counts"]
    assert j["fixed_length_input"]["identical_counts"]
    # r04: parity over every read with segments + a sample; the boundary sweep; the stress rows
    p = j["parity"]
    assert p["bit_exact"] and p["reads_with_segments"] == p["gpu_reads_with_segments"] > 5000 and p["checked_reads"] > 100_000
    ps = j["pack_sweep"]
    assert ps["threads"] == [1, 4, 8, 16] and 1000 in ps["pack_pairs"]
    row = ps["rows"]["gf_map_reads_hits, pageable"]["1000"]          # the reference's PACK_SIZE (common.rs:23)
    assert row[ps["threads"].index(8)] > 150.0 and row[0] > 40.0      # M reads/s: 8 threads, one thread
    assert all(v is not None and v <= 1000 for v in ps["smallest_pack_pairs_with_8_threads_over_50M_reads_per_s"].values())
    st = j["stress"]
    assert st["repeat30_reads_per_s"] > 0.6 * j["value"] and st["repeat30"]["high_keys"] > 10_000


def _tree_kernel_sha():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_sha()


@pytest.mark.parametrize("c,key", [(1, "IDX-D_20000000_150"), (2, "IDX-C_200000000_150"), (3, "IDX-D_200000000_150"),
                                   (4, "config4_50000000x16_150")])
def test_counter_entries_are_of_this_trees_kernels(c, key):
    """`frac` must not be a live time divided into a dead byte count."""
    j, t = _line(c), json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    e, r = t[key], j["roofline"]
    assert e["round"] == "r04" and e["git_head"] and e["git_dirty_csrc"] is False
    assert r["traffic_stale"] is False
    assert r["traffic_build"] == {"git_head": e["git_head"], "kernel_src_sha": e["kernel_src_sha"], "lib_sha": e["lib_sha"]}
    assert r["loaded_build"]["kernel_src_sha"] == e["kernel_src_sha"]
    # the tree's kernel sources are the profiled ones: touch genefuserust_amd/csrc or include/gfmatch.h and this fails
    # until tools/stamp_head.sh + tools/profile_all.sh have run again and the new lines are committed
    assert _tree_kernel_sha() == e["kernel_src_sha"], "kernel sources changed since the committed profile: profile again"
    assert r["hbm_only"] is None and "not separable" in r["hbm_only_note"]


@pytest.mark.parametrize("c,key", [(1, "IDX-D_20000000_150"), (2, "IDX-C_200000000_150"), (3, "IDX-D_200000000_150")])
def test_frac_recomputes_from_the_committed_profile(c, key):
    j, t = _line(c), json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    e, r = t[key], j["roofline"]
    assert e["source"].startswith("profiles/r04_cfg%d" % c) and os.path.exists(os.path.join(ROOT, e["source"]))
    assert r["traffic"] == e["hbm_bytes_per_launch"] == e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
    frac = e["hbm_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 8e12
    assert abs(frac - r["frac"]) < 1e-6
    # the per-kernel rows add up, and the calibration rows say what the correction rests on
    assert abs(sum(k["read_bytes"] + k["write_bytes"] for k in e["per_kernel"].values()) - e["hbm_bytes_per_launch"]) < 16
    if c == 1:
        cal = e["calibration"]["gf_k_calib_stream<0>"]
        assert abs(cal["dram_32B_units_x32_bytes"] - 3.0e9) < 1e6 and abs(cal["fetch_size_bytes_as_reported"] - 1.5e9) < 1e6
        # and the trace of the same run agrees with the line's kernel time within 5 %
        txt = open(os.path.join(ROOT, e["source"])).read()
        import re
        ms = sum(float(re.search(r"## %s[^\n]* dispatches: n=\d+ avg=([0-9.]+) ms" % re.escape(k), txt).group(1))
                 for k in e["per_kernel"])
        assert abs(ms - r["kernel_ms_avg"]) / r["kernel_ms_avg"] < 0.05
