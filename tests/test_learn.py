"""the learn loop (alphazero-risk_amd/learn.py).  CPU: host-side Game pieces against the oracle
(invertPlayers, trimOldExamples, acceptance rule).  GPU: one full tiny iteration end to end — self-play, train step,
new-vs-old arena through the batched Player seam, accept/revert, benchmark, reference log files and sample file."""
import argparse
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import ROOT  # noqa: F401

FM = T.data_field_mask()


def learn_mod():
    return importlib.import_module("alphazero-risk_amd.learn")


def test_invert_players_matches_oracle(orc):
    import host_arena as L
    g = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    states = g["states"][::13][:200].copy()
    got = L.invert_players(states)
    s = T.OrcState()
    want = np.zeros(160, np.uint8)
    for i in range(len(states)):
        orc.orc_state_unpack(C.byref(s), T.ptr(states[i]))
        orc.orc_invert_players(C.byref(s))
        orc.orc_state_pack(C.byref(s), T.ptr(want))
        assert (got[i][FM] == want[FM]).all()


def test_trim_and_acceptance_rules():
    L = learn_mod()
    rec = np.arange(100)[:, None].repeat(2, 1)
    # below the minimum: nothing is dropped (alphazero_nn_data.cpp:67-84)
    r, o = L.trim_old_examples(rec, 50, smin=200, smax=1000)
    assert len(r) == 100 and o == 50
    # above the minimum with old games present: drop min(oldGameIndex, excess) oldest
    r, o = L.trim_old_examples(rec, 30, smin=60, smax=1000)
    assert len(r) == 70 and o == 0 and r[0, 0] == 30
    r, o = L.trim_old_examples(rec, 50, smin=90, smax=1000)
    assert len(r) == 90 and o == 40
    # above the maximum: keep the newest smax
    r, o = L.trim_old_examples(rec, 10, smin=10, smax=40)
    assert len(r) == 40 and r[0, 0] == 60
    # isModelImproved: new.win >= (new.win + old.win) * 0.55, draws ignored (alphazero_trainer.cpp:192-198)
    assert L.is_model_improved(dict(win=[55, 45]), 0.55) and not L.is_model_improved(dict(win=[54, 46]), 0.55)
    assert L.is_model_improved(dict(win=[0, 0]), 0.55)
    assert L.gr_str(dict(draw=1, win=[2, 3], win_and_started=[1, 2])) == "1, 2/1, 3/2"


@pytest.mark.gpu
def test_one_learn_iteration_end_to_end(tmp_path):
    L = learn_mod()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        a = argparse.Namespace(ti=1, tg=8, mcts=6, gpu_games=8, blocks=1, e=2, bs=64, cg=4, ct=0.0, s=1024 * 512,
                               seed=77, dtype="bf16", device=0, include_compare_samples=1)
        out = L.learn(a, log=lambda *_: None)
    finally:
        os.chdir(cwd)
    assert len(out) == 1 and out[0]["improved"] and out[0]["samples"] > 500
    assert len(out[0]["losses"]) == 2 and out[0]["arena"]["count"] == 4
    for f in ("log/azr-improvement-log.txt", "log/azr-benchmark-log.txt", "log/azr-nn-training-log.txt",
              "checkpoints/latest-checkpoint.bin", "checkpoints/best-checkpoint.bin", "checkpoints/checkpoint-iter-0.bin",
              "data/training_samples.bin"):
        assert os.path.getsize(tmp_path / f) > 0, f
    from test_log_grammar import check   # the reference's own grammar (tests/golden/ref_logs.json) and log_chart.py's parsing
    imp_rows = check("improvement", open(tmp_path / "log/azr-improvement-log.txt").read())
    ben_rows = check("benchmark", open(tmp_path / "log/azr-benchmark-log.txt").read())
    nn_rows = check("nn", open(tmp_path / "log/azr-nn-training-log.txt").read())
    assert len(imp_rows) == 1 and imp_rows[0][0] == 0 and sum(imp_rows[0][1:3]) + imp_rows[0][4] == 4
    assert len(ben_rows) == 1 and len(nn_rows) == 1 and len(nn_rows[0]) == 4          # 2 epochs x (policy, value)
    imp = open(tmp_path / "log/azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and len(imp) == 4          # iter, draws, new W/Wstart, old W/Wstart
    bench = open(tmp_path / "log/azr-benchmark-log.txt").read().strip()
    assert bench.startswith("0,") and bench.count("/") == 4
    raw = open(tmp_path / "data/training_samples.bin", "rb").read()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    assert len(raw) == 8 + n * 265 and n == out[0]["samples"]


@pytest.mark.gpu
def test_host_stepped_arena_through_the_player_seam():
    """the batched Player seam of the binding (takeTurns = trim, simulate, argmax, makeMove until the turn passes) plays
    whole mirrored game pairs between two engines (tests/host_arena.py; the product's arena is the device one)"""
    import host_arena
    P = importlib.import_module("alphazero-risk_amd")
    a = P.Engine(4, blocks=1, sims=6, dtype=P.NET_BF16, max_game_rounds=20)
    b = P.Engine(4, blocks=1, sims=6, dtype=P.NET_BF16, max_game_rounds=20)
    a.init_random(1)
    b.init_random(2)
    r = host_arena.arena_two_nets(a, b, 8, True, 5)
    assert r["count"] == 8 and r["draw"] + r["win"][0] + r["win"][1] == 8
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dp", ["1", "0"])
def test_two_rank_learn_iteration(tmp_path, dp):
    """the N > 1 learn path (BASELINE configs[4]) rehearsed with 2 ranks on this box's one GPU over gloo: sharded
    self-play, all_gather of the records, training (--dp 1: data-parallel optimiser step on both ranks; --dp 0, the
    default: rank 0 trains and broadcasts the weights), sharded arena + benchmark with reduced
    GameResults; rank 0 writes the reference's files"""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, AZR_LEARN_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "alphazero-risk_amd", "learn.py"), "--ti", "1", "--tg", "8", "--mcts", "6",
           "--gpu-games", "8", "--blocks", "1", "-e", "2", "--bs", "64", "--cg", "8", "--ct", "0", "--dp", dp]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    assert "Model improved" in r.stdout and "Loss Policy / Value" in r.stdout
    imp = open(tmp_path / "log/azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and len(imp) == 4
    d, w0, w1 = int(imp[1]), int(imp[2].split("/")[0]), int(imp[3].split("/")[0])
    assert d + w0 + w1 == 8                                  # 2 ranks x 2 pairs
    bench = open(tmp_path / "log/azr-benchmark-log.txt").read().strip()
    nums = [int(x.split("/")[0]) for x in bench.replace(" ", "").split(",")[1:]]
    assert nums[0] + nums[1] + nums[2] == 10 and nums[3] + nums[4] + nums[5] == 100
    raw = open(tmp_path / "data/training_samples.bin", "rb").read()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    assert len(raw) == 8 + n * 265 and n > 500
