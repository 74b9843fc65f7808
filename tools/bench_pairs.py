#!/usr/bin/env python3
"""Pairs per second from FASTQ text resident in HBM to the hit list, on one MI355X: records
(gf_fastq_index_device + gf_fastq_gather_device, R1 and R2) -> ONE gf_scan_pairs_device call
(fast_merge, merged-or-R1+R2 mapping, reverse-complement retries, ordered compaction of the
matched reads: the policy of PairEndScanner::scan_pair_end, pescanner.rs:427-518) — and, as a
separate figure, the host-side tail (make_match / calc_distance per hit) that follows.

Synthetic pairs per SURVEY.md §8(d) (synth.make_pairs: fragments N(300,30) cut from the
druggable-shaped genes, PANEL mix, junction fragments planted).  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.bench_frontend import make_text, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--shape", default="IDX-D")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--check", type=int, default=3000)
    ap.add_argument("--profile-mode", action="store_true", help="the timed stages only (for rocprofv3): no host tail, no oracle")
    ap.add_argument("--full-gather", action="store_true",
                    help="copy the qualities out of the text too (gf_fastq_gather_device + gf_scan_pairs_device), as before r03 b")
    a = ap.parse_args()
    from genefuserust_amd import FusionMapper, Indexer, synth
    from genefuserust_amd.fastq import fastq_cut_device
    from genefuserust_amd.read_pair import finish_pair_hits, finish_pair_hits_device, scan_pairs_device
    dev = torch.device("cuda", 0)
    genes = synth.make_geneset(a.shape)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    n, L = a.pairs, a.read_len
    pr = synth.make_pairs(genes, n, read_len=L, seed=20240301, device="cuda")
    t1 = make_text(pr.l_bases, pr.l_quals, n, L, 1, dev)
    t2 = make_text(pr.r_bases, pr.r_quals, n, L, 2, dev)
    kinds = pr.kinds
    del pr
    torch.cuda.synchronize()
    lean = not a.full_gather   # the qualities stay in the text (gf_fastq_gather_lean_device, gf_scan_pairs_text_device)
    ms_cut1, b1 = timed(lambda: fastq_cut_device(ix, t1, lean=lean), a.steps, a.warmup)
    ms_cut2, b2 = timed(lambda: fastq_cut_device(ix, t2, lean=lean), a.steps, a.warmup)
    assert b1.n_records == b2.n_records == n and (b1.qual_off is not None) == lean
    ms_scan, res = timed(lambda: scan_pairs_device(ix, b1.bases, b1.quals, b1.offsets, b2.bases, b2.quals, b2.offsets, L,
                                                   l_qual_off=b1.qual_off, r_qual_off=b2.qual_off),
                         a.steps, a.warmup)
    ms_tail_dev, tail_dev = timed(lambda: finish_pair_hits_device(ix, res), a.steps, a.warmup)
    rec, hb, hq, tot = res.download()
    assert tot["overflow"] == 0, tot
    if a.profile_mode:
        print(json.dumps({"stage_ms": {"fastq_cut_R1": ms_cut1, "fastq_cut_R2": ms_cut2, "scan_pairs_device": ms_scan}, "totals": tot}))
        return 0
    fm = FusionMapper(ix)
    t0 = time.perf_counter()
    done = finish_pair_hits(fm, rec, hb, hq)
    tail_s = time.perf_counter() - t0
    # the library call alone (make_match + calc_distance per record on 8 host threads), without the Python objects
    from genefuserust_amd import _lib
    rm = np.zeros(rec.shape[0], dtype=_lib.READMATCH_DTYPE)
    st = np.zeros(rec.shape[0], dtype=np.int32)
    recc = np.ascontiguousarray(rec)
    dev_rm = tail_dev[0][: rec.shape[0]].cpu().numpy().view(_lib.READMATCH_DTYPE).reshape(-1)
    c_s = {}
    tail_equal = None
    for threads in (1, 8):
        t0 = time.perf_counter()
        for _ in range(5):
            _lib.check(_lib.lib().gf_pair_hits_finish(ix._handle(), recc.ctypes.data, rec.shape[0], hb, len(hb), rm.ctypes.data,
                                                      st.ctypes.data, threads))
        c_s[threads] = (time.perf_counter() - t0) / 5
    tail_equal = bool(dev_rm.tobytes() == rm.tobytes())
    # parity sample: the records of the first pairs against the oracle-driven policy
    from oracle import oracle_py
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from tests.test_pair_pipeline import _reference_policy
    ox = oracle_py.OracleIndexer(genes.seqs)
    # sample: the first pairs of the pack and the pairs of the first hit records (so that matches are checked)
    ids = sorted(set(range(min(a.check, n))) | {int(x) for x in rec["pair_id"][:300]})
    o1, o2 = b1.offsets.cpu().numpy(), b2.offsets.cpu().numpy()
    sel = torch.tensor(ids, device=dev)

    def rows(t, off):
        return [t[int(off[i]):int(off[i + 1])].cpu().numpy().tobytes() for i in ids]

    def qrows(b, off):   # the qualities of the sampled records: gathered, or still in the text (lean cut)
        if b.qual_off is None:
            return rows(b.quals, off)
        qo = b.qual_off.cpu().numpy()
        return [b.quals[int(qo[i]):int(qo[i]) + int(off[i + 1] - off[i])].cpu().numpy().tobytes() for i in ids]
    L1b, L1q, L2b, L2q = rows(b1.bases, o1), qrows(b1, o1), rows(b2.bases, o2), qrows(b2, o2)
    pairs = list(zip(L1b, L1q, L2b, L2q))
    want, _ = _reference_policy(oracle_py, ox, genes.reversed_flags, pairs)
    flat = [(ids[p], w) for p, ws in enumerate(want) for w in ws]
    idset = set(ids)
    got = [h for h in rec if int(h["pair_id"]) in idset]
    bad = int(len(got) != len(flat))
    for h, (p, (source, on_rc, m_rev, seq, qual, rm)) in zip(got, flat):
        o, ln = int(h["seq_offset"]), int(h["read_len"])
        bad += not (int(h["pair_id"]) == p and int(h["source"]) == source and bool(h["flags"] & 1) == on_rc and
                    hb[o:o + ln] == seq and hq[o:o + ln] == qual)
    k = len(ids)
    total = ms_cut1 + ms_cut2 + ms_scan
    text_bytes = int(t1.numel() + t2.numel())
    # roofline per stage: counter-measured bytes (tools/profile_pairs.sh -> profiles/frontend_traffic.json, calibrated as
    # in NOTEBOOK.md §5 r03: read = 32 B x TCC_EA0_RDREQ_DRAM_32B, written = WRITE_SIZE) over this run's stage time, and the
    # stage's algorithmic in + out bytes (text in; bases, qualities, offsets out / records in, hit records out)
    roof = {}
    try:
        ft = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "frontend_traffic.json")))
        scale = n / ft["pairs"]
        rec_bytes = int(t1.numel()) // n
        # (lean: a cut writes no qualities but one quality offset per record; the scan reads no qualities to speak of)
        algo = {"fastq_cut": rec_bytes * n + (L * n + 8 * n if lean else 2 * L * n) + 8 * (n + 1),
                "scan_pairs_device": 2 * ((L if lean else 2 * L) * n + 8 * (n + 1)) + 64 * int(tot["hits"]) + 2 * int(tot["hit_bytes"])}
        if bool(ft.get("lean")) != lean:
            raise ValueError("profiles/frontend_traffic.json was measured %s the qualities' copy" % ("without" if ft.get("lean") else "with"))
        for st, ms in (("fastq_cut", 0.5 * (ms_cut1 + ms_cut2)), ("scan_pairs_device", ms_scan)):
            e = ft["stages"][st]
            b = (e["read_bytes"] + e["write_bytes"]) * scale
            roof[st] = {"bound": "hbm", "ms": ms, "traffic": int(b), "achieved": b / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                        "frac": b / ms / 1e6 / 8000.0, "frac_of_copy_ceiling_6290": b / ms / 1e6 / 6290.0,
                        "algorithmic_bytes": algo[st], "algorithmic_GBps": algo[st] / ms / 1e6,
                        "algorithmic_frac": algo[st] / ms / 1e6 / 8000.0,
                        "kernel_ms_when_profiled": e["kernel_ms"], "traffic_source": "profiles/frontend_traffic.json (%s)" % ft.get("source_dir")}
    except Exception as e:  # noqa: BLE001
        roof = {"error": "%s: %s" % (type(e).__name__, e)}
    print(json.dumps({
        "metric": "read pairs per second from FASTQ text in HBM to the hit list (records + scan_pair_end policy in one device call)",
        "value": n / (total / 1e3), "unit": "pairs/s", "pairs": n, "read_len": L, "index_shape": a.shape,
        "stage_ms": {"fastq_cut_R1": round(ms_cut1, 3), "fastq_cut_R2": round(ms_cut2, 3), "scan_pairs_device": round(ms_scan, 3)},
        "scan_pairs_only_pairs_per_s": n / (ms_scan / 1e3),
        "qualities": ("left in the text: gf_fastq_gather_lean_device + gf_scan_pairs_text_device" if lean else
                      "copied out of the text: gf_fastq_gather_device + gf_scan_pairs_device"),
        "roofline": roof,
        "totals": tot, "junction_pairs": int((kinds == 2).sum()),
        "host_tail": {"hits": len(done), "seconds": round(tail_s, 4), "hits_per_s": len(done) / tail_s if tail_s else None,
                      "note": "finish_pair_hits: gf_pair_hits_finish + one Python ReadMatch object per hit",
                      "library_call_seconds": {str(k): round(v, 5) for k, v in c_s.items()},
                      "library_hits_per_s_8_threads": rec.shape[0] / c_s[8] if c_s[8] else None,
                      "device_tail_ms": round(ms_tail_dev, 4),
                      "device_tail_hits_per_s": rec.shape[0] / (ms_tail_dev / 1e3) if ms_tail_dev else None,
                      "device_tail_equals_host": tail_equal},
        "text_bytes": text_bytes, "text_GBps": text_bytes / (total / 1e3) / 1e9,
        "parity": {"checked_pairs": k, "records_expected": len(flat), "mismatches": bad}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
