"""GPU: the RCCL ("nccl") branch of the multi-rank code, executed on ONE MI355X.

BASELINE configs[3] / configs[4] shard over 8 GPUs; a gpurun box has one.  What can and must run here is the CODE that only
executes under backend "nccl": device-tensor all_gather of the (s, pi, z) records, device-tensor all_reduce of counters,
the weight broadcast, and the data-parallel optimiser step whose all-reduce callback hands RCCL an alias of an engine
buffer (`__cuda_array_interface__`).  AZR_FORCE_DIST=1 makes a world of one rank take every collective instead of the
single-process shortcuts (alphazero-risk_amd/shard.py:force_dist); with one rank each collective is the identity, so the
results must equal the single-process ones.  Reference structure replaced: one self-play thread per GPU + vector concat
(alphazero_trainer.cpp:41-62), weight hand-over through temp.bin (alphazero_gpu_cluster.cpp:221-231)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import ROOT

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **kw)
    return e


def test_bench_record_exchange_over_rccl_with_one_rank():
    """bench.py's exchange leg on real finished records: counts + padded all_gather of DEVICE tensors over backend nccl"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--games", "32", "--sims", "8", "--blocks", "1", "--steps", "2",
           "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--tail-seconds", "30"]
    r = subprocess.run(cmd, env=_env(AZR_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ex = out["exchange"]
    assert "backend nccl" in ex["collective"] and ex["gathered_on"].startswith("cuda")
    assert ex["records_gathered"] == ex["records_this_rank"] > 0 and ex["bytes"] == 265 * ex["records_gathered"]
    assert out["n_gpus"] == 1 and out["records_dropped"] == 0 and out["errors"] == 0 and out["value"] > 0
    # the same run without the collectives gathers the same number of records (same seeds, same passes)
    r1 = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    o1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert o1["exchange"]["collective"].startswith("none") and o1["exchange"]["records_gathered"] > 0


@pytest.mark.parametrize("dp", ["0", "1"])
def test_learn_iteration_over_rccl_with_one_rank(tmp_path, dp):
    """one toy learn iteration with cdev = cuda: record all_gather, counter all_reduce, and either the weight broadcast
    (--dp 0, the reference's GPU-0-trains structure) or the data-parallel optimiser step with RCCL all-reduces on the
    engine's own device buffers (--dp 1)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "alphazero-risk_amd", "learn.py"), "--ti", "1", "--tg", "8", "--mcts", "6",
           "--gpu-games", "8", "--blocks", "1", "-e", "2", "--bs", "64", "--cg", "8", "--ct", "0", "--dp", dp, "--dp-callback", "1"]
    r = subprocess.run(cmd, cwd=tmp_path, env=_env(AZR_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    assert "Record exchange: all_gather + counter all_reduce, backend nccl, tensors on cuda:0" in r.stdout
    if dp == "1":
        assert "Data-parallel optimiser step: torch.distributed all_reduce on device buffers, backend nccl" in r.stdout
    else:
        assert "Weight broadcast from rank 0: backend nccl, tensor on cuda:0" in r.stdout
    assert "Model improved" in r.stdout and "Loss Policy / Value" in r.stdout
    imp = open(tmp_path / "log/azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and int(imp[1]) + int(imp[2].split("/")[0]) + int(imp[3].split("/")[0]) == 8
    raw = open(tmp_path / "data/training_samples.bin", "rb").read()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    assert len(raw) == 8 + n * 265 and n > 500


def test_data_parallel_step_over_rccl_aliases_engine_buffers(tmp_path):
    """azr_nn_train_dp with make_allreduce(dist, on_device=True): every all-reduce (float64 batch-norm sums, float32 losses,
    the whole float32 gradient vector) is an RCCL collective on a torch tensor that ALIASES the engine's buffer
    (asserted inside make_allreduce: same pointer, same device).  One rank: sums are the identity, so the step must
    equal the single-GPU azr_nn_train step like the 2- and 4-rank gloo runs do."""
    blocks = 2
    out = str(tmp_path / "dp.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "helpers", "dp_train_worker.py"), out]
    r = subprocess.run(cmd, env=_env(DP_BLOCKS=str(blocks), DP_BS="64", DP_N="200", DP_BACKEND="nccl"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    d = np.load(out)
    assert str(d["backend"]) == "nccl" and int(d["world"]) == 1
    L = 2 * blocks + 1
    per_step = 2 * (L + 1) + 2
    for tag, steps in (("one", 1), ("multi", 6)):
        calls = d[f"{tag}_calls"]
        assert len(calls) == steps * per_step, (len(calls), per_step)   # the callback really ran at world = 1
        assert (calls[:, 0] == len(d["w0"])).sum() == steps
        assert set(calls[:, 1].tolist()) == {0, 1}                      # float32 and float64 buffers both went through RCCL
        assert int(d[f"{tag}_state_dp"]) == int(d[f"{tag}_state_1"])
    assert np.abs(d["one_hist_dp"] - d["one_hist_1"]).max() <= 2e-5
    worst = 0.0
    for name, off, n in T.net_layout(blocks):
        a, b = d["one_g_dp"][off:off + n], d["one_g_1"][off:off + n]
        if name.endswith("_bn"):
            a, b = a[:n // 2], b[:n // 2]
        scale = np.abs(b).max()
        if scale > 0:
            worst = max(worst, np.abs(a - b).max() / scale)
    assert worst <= 2e-3, worst
    dw = np.abs(d["one_w_dp"] - d["one_w_1"])
    moved = np.abs(d["one_w_1"] - d["w0"]) > 0
    assert np.median(dw[moved]) <= 1e-7 and (dw[moved] > 1e-5).mean() <= 2e-3
    rel = np.abs(d["multi_hist_dp"] - d["multi_hist_1"]) / np.abs(d["multi_hist_1"])
    assert rel.max() <= 1e-3
    print(f"RCCL, one rank: max rel grad diff {worst:.2e}, median |dw| {np.median(dw[moved]):.1e}, multi-step loss rel diff {rel.max():.1e}")


def test_data_parallel_step_on_the_engines_own_rccl_communicator(tmp_path):
    """azr_dp_init + azr_nn_train_dp(allreduce = NULL): the handle's own communicator (RCCL bound at run time), every sum an
    ncclAllReduce on the engine's stream.  One rank (one GPU here): each sum is the identity, so the step equals the single-GPU step
    like the callback form; the id travels through torch.distributed as in learn.py."""
    blocks = 2
    out = str(tmp_path / "dp.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "helpers", "dp_train_worker.py"), out]
    r = subprocess.run(cmd, env=_env(DP_BLOCKS=str(blocks), DP_BS="64", DP_N="200", DP_BACKEND="nccl", DP_NATIVE="1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    d = np.load(out)
    assert len(d["one_calls"]) == 0 and len(d["multi_calls"]) == 0     # no callback ran: the sums went through the communicator
    assert int(d["one_state_dp"]) == int(d["one_state_1"])
    assert np.abs(d["one_hist_dp"] - d["one_hist_1"]).max() <= 2e-5
    worst = 0.0
    for name, off, n in T.net_layout(blocks):
        a, b = d["one_g_dp"][off:off + n], d["one_g_1"][off:off + n]
        if name.endswith("_bn"):
            a, b = a[:n // 2], b[:n // 2]
        scale = np.abs(b).max()
        if scale > 0:
            worst = max(worst, np.abs(a - b).max() / scale)
    assert worst <= 2e-3, worst
    rel = np.abs(d["multi_hist_dp"] - d["multi_hist_1"]) / np.abs(d["multi_hist_1"])
    assert rel.max() <= 1e-3
    assert np.abs(d["multi_w_dp"] - d["w0"]).max() > 1e-3
    print(f"native RCCL communicator, one rank: max rel grad diff {worst:.2e}, multi-step loss rel diff {rel.max():.1e}")


def test_learn_iteration_with_the_native_communicator(tmp_path):
    """learn.py --dp 1 on device tensors takes the engine's own communicator by default (the torch.distributed callback only with
    --dp-callback 1)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "alphazero-risk_amd", "learn.py"), "--ti", "1", "--tg", "8", "--mcts", "6",
           "--gpu-games", "8", "--blocks", "1", "-e", "2", "--bs", "64", "--cg", "8", "--ct", "0", "--dp", "1"]
    r = subprocess.run(cmd, cwd=tmp_path, env=_env(AZR_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    assert "Data-parallel optimiser step: in-stream ncclAllReduce on the engine's own communicator" in r.stdout
    assert "Model improved" in r.stdout and "Loss Policy / Value" in r.stdout
