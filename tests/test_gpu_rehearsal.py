"""GPU: BASELINE configs[3] / configs[4] rehearsed at their real rank counts and shapes on the ONE MI355X of a gpurun box.

A box allows at most 6 processes on its card (the test runner is one of them), so:
  * the data-parallel optimiser step at WORLD 8 and the reference's BATCH_SIZE 512 (64-record shares -> the small-batch conv kernel
    t_conv_q, 8-board weight-gradient slices) runs as 8 engine handles driven by 8 host threads of THIS process, the sums meeting in an
    in-process all-reduce (rank order, so the ranks must end bit-equal) — the C-ABI's "one handle = one host thread" contract at 8
    handles on one device is exactly what the C++ host's one-thread-per-GPU structure relies on;
  * the multi-PROCESS paths run over gloo with the ranks sharing the card, at the largest world that fits beside the runner: bench.py's
    own rank spawner (its parent never touches HIP) + record gather with 5 ranks, one learn.py iteration under torchrun (whose agent
    process opens the GPU too) with 4 — ragged game / pair / record counts in both;
  * the C++ host CLI's in-process multi-GPU path (one host thread per GPU, temp.bin weight hand-over, split arena, merged results)
    runs with --gpus 2 and both logical GPUs mapped onto device 0 (--devices 0,0).
Reference structures replaced: one self-play thread per GPU + vector concat (alphazero_trainer.cpp:41-62), GPU-0-trains + temp.bin
(alphazero_gpu_cluster.cpp:144-164,221-231), GameGroup's thread-per-pair fan-out (game.cpp:277-312)."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import ROOT, pkg

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "alphazero-risk_amd", "host")
EXE = os.path.join(HOST, "AlphaZero_Risk_hip")


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **kw)
    return e


def _records(n, seed):
    rng = np.random.default_rng(seed)
    in88 = np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"]
    rec = np.zeros((n, 265), np.uint8)
    rec[:, 0] = rng.integers(0, 2, n)
    rec[:, 1:89] = in88[rng.integers(0, len(in88), n)]
    rec[:, 89:93] = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), n).view(np.uint8).reshape(n, 4)
    pi = rng.random((n, 43)).astype(np.float32) ** 3
    pi /= pi.sum(1, keepdims=True)
    rec[:, 93:265] = pi.view(np.uint8).reshape(n, 172)
    return rec


class ThreadAllReduce:
    """an all-reduce among `world` host threads of one process: every rank copies its device buffer to the host, all meet, every rank
    adds the contributions in RANK ORDER (so all ranks hold the same bits) and writes the sum back"""

    def __init__(self, world, device=0):
        import torch
        self.torch, self.world = torch, world
        self.dev = torch.device("cuda", device)
        self.slots = [None] * world
        self.bar = threading.Barrier(world, timeout=300)
        self.calls = [[] for _ in range(world)]

    def make(self, rank):
        torch = self.torch
        shard = __import__("importlib").import_module("alphazero-risk_amd.shard")

        def ar(ptr, count, dtype):
            self.calls[rank].append((count, dtype))
            with torch.cuda.device(self.dev):
                t = torch.as_tensor(shard._DevicePtr(ptr, count, dtype), device=self.dev)
                assert t.data_ptr() == int(ptr)
                self.slots[rank] = t.cpu()
                self.bar.wait()
                tot = self.slots[0].clone()
                for k in range(1, self.world):
                    tot += self.slots[k]
                self.bar.wait()              # everybody has read every slot before anybody overwrites its own
                t.copy_(tot)
                torch.cuda.synchronize(self.dev)
        return ar


def test_world8_batch512_data_parallel_step_in_one_process():
    """configs[4]'s optimiser step at its real shape: 8 ranks, BATCH_SIZE 512 (settings.h:74) -> 64-record shares, 20-block layout
    rules at B = 2: two steps; ranks bit-equal, losses / gradients / weights against the single-GPU step of the same minibatches"""
    import torch
    torch.cuda.init()
    P = pkg()
    world, bs, blocks = 8, 512, 2
    flat = T.make_net_flat(blocks, seed=9, perturb_bn=True)
    rec = _records(2 * bs, seed=5)
    ar = ThreadAllReduce(world)
    engs = [P.Engine(4, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64) for _ in range(world)]
    for e in engs:
        e.set_weights(flat)
    ref = P.Engine(4, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    ref.set_weights(flat)
    L = 2 * blocks + 1
    per_step = 2 * (L + 1) + 2
    state = state1 = 4321
    for step in range(2):   # two calls of one step each: gradients are compared after the first, the Adam state carries into the second
        mb = rec[step * bs:(step + 1) * bs]
        out, errs = [None] * world, [None] * world

        def run(r):
            try:
                out[r] = engs[r].train_dp(mb, 1, ar.make(r), r, world, batch_size=bs, rng_state=state)
            except BaseException as ex:   # noqa: BLE001
                errs[r] = ex
                ar.bar.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
        assert not any(t.is_alive() for t in th) and errs == [None] * world, errs
        w = [e.get_weights() for e in engs]
        hist1, state1 = ref.train(mb, 1, batch_size=bs, rng_state=state1)
        w1 = ref.get_weights()
        for r in range(world):
            assert (w[r].view(np.uint32) == w[0].view(np.uint32)).all(), (step, r)   # same reduced sums, same Adam step: bit-equal ranks
            assert out[r] == out[0] and out[r][1] == state1, (step, r)                # same losses, same shuffle stream consumed
            assert len(ar.calls[r]) == (step + 1) * per_step and sum(c == len(flat) for c, _ in ar.calls[r]) == step + 1
        state = out[0][1]
        assert np.abs(np.array(out[0][0]) - np.array(hist1)).max() <= 2e-5 * (1 + 9 * step), (step, out[0][0], hist1)
        if step == 0:
            # the step's gradients, tensor by tensor.  At 512 records a ReLU input sits within fp32 rounding of zero somewhere in the
            # batch, and a mask flipped by a different summation order is a discrete change of single gradient entries: the measure is
            # the one of tests/test_gpu_train.py's batch-512 case, the relative L2 error per tensor
            g, g1 = engs[0].train_grads(), ref.train_grads()
            worst = 0.0
            for name, off, n in T.net_layout(blocks):
                a, b = g[off:off + n].astype(np.float64), g1[off:off + n].astype(np.float64)
                if name.endswith("_bn"):
                    a, b = a[:n // 2], b[:n // 2]
                nb = np.linalg.norm(b)
                if nb > 0:
                    worst = max(worst, np.linalg.norm(a - b) / nb)
            assert worst <= 3e-3, worst
            dw = np.abs(w[0] - w1)
            moved = np.abs(w1 - flat) > 0
            assert np.median(dw[moved]) <= 2e-7 and (dw[moved] > 2e-5).mean() <= 5e-3, (np.median(dw[moved]), (dw[moved] > 2e-5).mean())
            print(f"world 8 x 64 records: worst per-tensor relative L2 gradient error {worst:.2e}, median |dw| {np.median(dw[moved]):.1e}, "
                  f"{per_step} all-reduces per step")
    for e in engs:
        e.close()
    ref.close()


def test_bench_five_ranks_over_gloo():
    """`python bench.py --gpus 5` (its own spawner: a parent that never touches HIP starts the ranks) with the gloo rehearsal backend:
    5 disjoint seed streams, weak-scaling totals, the padded record gather of 5 ragged counts"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--games", "16", "--sims", "8", "--blocks", "1", "--steps", "2",
           "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--tail-seconds", "40"]
    r = subprocess.run(cmd, env=_env(AZR_BENCH_BACKEND="gloo", AZR_BENCH_DEADLINE_S="600"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 5 and out["scaling"] == "weak" and out["value"] > 0 and out["errors"] == 0 and out["records_dropped"] == 0
    ex = out["exchange"]
    assert "backend gloo" in ex["collective"] and ex["records_gathered"] > ex["records_this_rank"] > 0
    assert ex["bytes"] == 265 * ex["records_gathered"]
    # every rank ran its own games and contributed its records
    assert ex["records_gathered"] >= 3 * ex["records_this_rank"] and 0.9 <= out["decisions_per_game_and_step"] <= 2.0, out


@pytest.mark.parametrize("dp", ["1", "0"])
def test_learn_iteration_four_ranks_ragged_over_gloo(tmp_path, dp):
    """one learn.py iteration at world 4 over gloo (ranks share the card; with the runner and torchrun's agent that is the box's 6
    processes): 7 self-play games over 4 ranks (2, 2, 2, 1), record gather of 4 ragged counts, training — the data-parallel step with
    16-record shares of a 64-record minibatch (--dp 1) or rank 0 + weight broadcast (--dp 0, the default) —, 3 compare pairs over 4
    ranks (one rank plays nothing and contributes an EMPTY record shard), 5 + 50 benchmark pairs split, GameResults reduced, rank 0
    writes the reference's files"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "alphazero-risk_amd", "learn.py"), "--ti", "1", "--tg", "7", "--mcts", "6",
           "--gpu-games", "8", "--blocks", "1", "-e", "2", "--bs", "64", "--cg", "6", "--ct", "0", "--dp", dp, "--phase-deadline", "600"]
    r = subprocess.run(cmd, cwd=tmp_path, env=_env(AZR_LEARN_BACKEND="gloo"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    assert "world 4" in r.stdout and "Model improved" in r.stdout and "Loss Policy / Value" in r.stdout
    assert ("Data-parallel optimiser step" in r.stdout) == (dp == "1")
    assert ("Weight broadcast from rank 0" in r.stdout) == (dp == "0")
    assert "[7 games," in r.stdout
    imp = open(tmp_path / "log/azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and int(imp[1]) + int(imp[2].split("/")[0]) + int(imp[3].split("/")[0]) == 6
    bench = open(tmp_path / "log/azr-benchmark-log.txt").read().strip()
    nums = [int(x.split("/")[0]) for x in bench.replace(" ", "").split(",")[1:]]
    assert nums[0] + nums[1] + nums[2] == 10 and nums[3] + nums[4] + nums[5] == 100
    raw = open(tmp_path / "data/training_samples.bin", "rb").read()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    assert len(raw) == 8 + n * 265 and n > 500


def _host_exe():
    subprocess.check_call(["make", "-s", "-C", HOST])
    return EXE


def test_host_cli_train_two_gpus_on_one_card(tmp_path):
    """`AlphaZero_Risk_hip -m train --gpus 2 --devices 0,0 --ti 1`: the C++ host's in-process multi-GPU path — one host thread per GPU
    for self-play (disjoint seed streams, storages concatenated in GPU order), GPU 0 trains and hands the weights to GPU 1 through
    checkpoints/temp.bin, the compare pairs and the benchmark pairs split over both GPUs, results merged"""
    exe = _host_exe()
    cmd = [exe, "-m", "train", "--gpus", "2", "--devices", "0,0", "--ti", "1", "--tg", "9", "--mcts", "6", "--gpu-games", "8", "--blocks", "1",
           "-e", "1", "--bs", "64", "--cg", "10", "--ct", "0", "-t", "2"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    out = r.stdout
    assert "GPUs: 2" in out and "Model improved" in out
    assert "[gpu 0] games 5/5" in out and "[gpu 1] games 4/4" in out          # 9 games: 5 + 4
    assert "Self-play: 9 games" in out
    assert os.path.exists(tmp_path / "checkpoints/temp.bin")                      # AlphaZeroNNGroup::train's hand-over file
    imp = open(tmp_path / "log/azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and int(imp[1]) + int(imp[2].split("/")[0]) + int(imp[3].split("/")[0]) == 10   # 5 pairs: 3 + 2
    bench = open(tmp_path / "log/azr-benchmark-log.txt").read().strip()
    nums = [int(x.split("/")[0]) for x in bench.replace(" ", "").split(",")[1:]]
    assert nums[0] + nums[1] + nums[2] == 10 and nums[3] + nums[4] + nums[5] == 100
    # GPU 0's trained weights = the hand-over file = the accepted checkpoint every GPU of the generate group then loaded
    assert open(tmp_path / "checkpoints/best-checkpoint.bin", "rb").read() == open(tmp_path / "checkpoints/temp.bin", "rb").read()


def test_host_cli_play_two_gpus_on_one_card(tmp_path):
    """`-m play --gpus 2 --devices 0,0`: the game quota split in whole pairs over two host threads, results merged; an engine that
    cannot be created ends the CLI with exit code 1 and the engine's message (failures INSIDE the per-GPU threads:
    tests/test_host_threads.py)"""
    exe = _host_exe()
    r = subprocess.run([exe, "-m", "play", "--gpus", "2", "--devices", "0,0", "--mcts=8", "--cg=12", "--blocks=1", "--gpu-games", "4"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    tail = r.stdout.strip().split("\n")[-4:]
    assert tail[0] == "Games: 12" and sum(int(t.split(":")[1]) for t in tail[1:]) == 12
    bad = subprocess.run([exe, "-m", "play", "--gpus", "2", "--devices", "0,99", "--mcts=8", "--cg=4", "--blocks=1"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert bad.returncode == 1 and "fatal:" in bad.stderr and "terminate" not in bad.stderr, (bad.returncode, bad.stderr[-500:])
