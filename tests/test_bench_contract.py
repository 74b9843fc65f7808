"""The bench line's contract fields, checked on the CPU against the committed r03 lines (profiles/r03_bench_config*.json are
what bench.py printed on the MI355X): metric / unit / scaling / roofline / cpu_baseline shape, and that `roofline.frac`
can be recomputed from profiles/hbm_traffic.json and the line's own kernel time — what VERDICT r02 asked to be able to do."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(c):
    return json.load(open(os.path.join(ROOT, "profiles", "r03_bench_config%d.json" % c)))


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
    assert r["bound"] == "l2_tag" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 0.9
    assert j["parity"] and all(v["bit_exact"] for v in (j["parity"].values() if c == 4 else [j["parity"]]))


def test_config1_line_carries_cpu_baseline_and_extras():
    j = _line(1)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "reads/s" and cb["value"] > 1e5 and cb["sample"]
    assert "pairs" in j["config"]["workload"] and "N(300,30)" in j["config"]["workload"]
    assert j["h2d_inclusive"]["packed"]["same_hits_as_ascii"] and j["packed_input"]["identical_counts"]
    assert j["fixed_length_input"]["identical_counts"]


@pytest.mark.parametrize("c,key", [(1, "IDX-D_20000000_150"), (2, "IDX-C_200000000_150"), (3, "IDX-D_200000000_150")])
def test_frac_recomputes_from_the_committed_profile(c, key):
    j, t = _line(c), json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    e, r = t[key], j["roofline"]
    assert e["source"].startswith("profiles/r03_cfg%d" % c) and os.path.exists(os.path.join(ROOT, e["source"]))
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
