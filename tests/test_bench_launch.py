"""bench.py --gpus N run plainly (no WORLD_SIZE): it must start its N ranks itself, as a child process before
anything touches the GPU, relay rank 0's single JSON line and exit with the child's code (a driver that
uses the command shape of N = 1 got `SystemExit("launch with torch.distributed.run")` and no SCALE line)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_plain_gpus_n_starts_its_ranks_itself_cpu():
    """Without a GPU every rank stops at 'bench.py needs a GPU': seeing that message from the ranks, and a
    non-zero exit code, shows the launcher ran and its code came back."""
    if torch.cuda.is_available():
        pytest.skip("covered by the rehearsal test on the GPU box")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=_env())
    assert out.returncode != 0
    assert "bench.py needs a GPU" in out.stderr, out.stderr[-2000:]
    assert "launch with torch.distributed.run" not in out.stderr


@pytest.mark.gpu
def test_plain_gpus_2_prints_one_line_in_rehearsal_mode(gpu_device):
    """Two ranks on the box's one GPU (gloo exchange: RCCL refuses two ranks on one device): the N > 1 code path,
    launched by the plain command."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "200000", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=_env(GF_BENCH_REHEARSAL="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["parity"]["bit_exact"]
    assert j["h2d_inclusive"]["reads_per_s_all_ranks"] > 0 and len(j["h2d_inclusive"]["per_rank_reads_per_s"]) == 2


def _one_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_rehearsal_four_ranks_configs_1_3_4(gpu_device):
    """The N > 1 code paths of configs 1, 3 and 4 with FOUR ranks on the box's one GPU (gloo exchange; sized down).  Four,
    not eight: a GPU box admits at most 6 processes on its card at once, and the test runner and the launcher's agent
    hold it too (six ranks were killed by the box's process guard, r04) — the 8-rank plan and its group exchange are
    covered on the CPU (tests/test_dist.py::test_multi_csv_groups_of_four_ranks_gloo).  Never a measurement: the first
    real multi-GPU run is the driver's (fusion_scan.rs:103-116 is the split being rehearsed)."""
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-h2d", "--no-pack-sweep", "--no-stress"]
    env = _env(GF_BENCH_REHEARSAL="1")
    # config 1: weak scaling, every rank its own shard
    j = _one_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--pairs", "60000"] + common,
                                 capture_output=True, text=True, timeout=900, env=env))
    assert j["n_gpus"] == 4 and j["scaling"] == "weak" and j["config"]["ranks_seen"] == 4 and j["parity"]["bit_exact"]
    assert j["config"]["reads_per_gpu_per_step"] == 120000 and j["config"]["hits_per_step"] > 0
    # config 3: strong scaling, one batch sharded over the ranks
    j = _one_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "3", "--pairs", "200000"] + common,
                                 capture_output=True, text=True, timeout=900, env=env))
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["config"]["reads_per_gpu_per_step"] == 100000
    assert j["parity"]["bit_exact"] and j["config"]["ranks_seen"] == 4
    # config 4: 5 CSVs over 4 ranks (rank 0 owns two), then 2 CSVs over 4 ranks (groups of two, one exchange per CSV)
    for n_csv in (5, 2):
        j = _one_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "4", "--pairs", "60000",
                                      "--n-csv", str(n_csv), "--scale", "0.05"] + common,
                                     capture_output=True, text=True, timeout=900, env=env))
        assert j["n_gpus"] == 4 and j["config"]["n_csv"] == n_csv
        assert all(v["bit_exact"] for v in j["parity"].values())
