"""bench.py's launch contract: --gpus N is honoured (N ranks are really started, or the run fails loudly), never
silently ignored.  The CPU half runs here; the GPU half is a 2-rank gloo rehearsal on one card."""
import json
import os
import subprocess
import sys

import pytest

from gpu_common import ROOT

BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--games", "32", "--sims", "8", "--blocks", "1", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline",
         "--tail-seconds", "30"]


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_world_size_mismatch_fails_before_anything_runs():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1" in r.stderr


def test_gpus_flag_spawns_ranks_and_propagates_their_failure():
    """without a GPU every rank refuses to run (no CPU fallback); the parent must report that, not print a number"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_two_rank_bench_on_one_gpu")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr and "needs an MI355X" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_two_rank_bench_on_one_gpu():
    """plain `python bench.py --gpus 2` (no launcher): two ranks, n_gpus 2, twice the work of one rank, and a record
    exchange that really moved records (gloo backend: both ranks share this box's one card)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_env(AZR_BENCH_BACKEND="gloo"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["games_per_gpu"] == 32 and "32 concurrent" in out["config"]["workload"]
    ex = out["exchange"]
    assert ex["records_gathered"] > ex["records_this_rank"] > 0 and ex["bytes"] == 265 * ex["records_gathered"]
    assert out["records_dropped"] == 0 and out["errors"] == 0
    one = subprocess.run([sys.executable, BENCH, "--gpus", "1"] + SMALL, env=_env(), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert o1["n_gpus"] == 1 and o1["exchange"]["records_gathered"] > 0
    assert out["roofline"]["leaf_slots_per_launch"] == o1["roofline"]["leaf_slots_per_launch"] == 64


@pytest.mark.gpu
def test_bench_under_torchrun_as_the_driver_launches_it():
    """the driver's own form for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N (gloo here:
    the two ranks share this box's one card)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, env=_env(AZR_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["exchange"]["records_gathered"] > out["exchange"]["records_this_rank"] > 0
    assert "extra_configs" not in out and "cpu_baseline" not in out      # N > 1: the headline only
