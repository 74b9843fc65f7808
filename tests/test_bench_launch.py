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
