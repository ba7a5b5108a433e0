"""The C++ host side (alphazero-risk_amd/host): reference-shaped Settings/CLI.  CPU part: flag parsing, defaults,
log/settings.txt side effect, loud failure without a device.  GPU part: `-m learn` and `-m play` end to end."""
import os
import subprocess

import numpy as np
import pytest

from gpu_common import ROOT, have_gpu

HOST = os.path.join(ROOT, "alphazero-risk_amd", "host")
EXE = os.path.join(HOST, "AlphaZero_Risk_hip")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "alphazero-risk_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", HOST])
    return EXE


def test_help_lists_every_reference_flag(exe):
    out = subprocess.run([exe, "--help"], capture_output=True, text=True, check=True).stdout
    for flag in ("-m", "-g", "-c", "--p1", "--g1", "--c1", "--p2", "--g2", "--c2", "--gpus", "--gpu-games", "-t", "--apbs",
                 "--lnt", "--ls", "--dgss", "--dgsr", "--dtl", "--allow-yield", "--limit-reinforcement", "--limit-attack",
                 "--mirror-games", "--ti", "--tg", "--mcts", "--hp", "--dnv", "--dne", "--temp", "-e", "--bs", "--cg",
                 "--ct", "-s"):
        assert flag.lstrip("-") in out, flag


def test_settings_file_and_derived_gpu_games(exe, tmp_path):
    if have_gpu():
        pytest.skip("needs a box without a GPU (checks the loud failure)")
    r = subprocess.run([exe, "-m", "learn", "--mcts=16", "-t", "4", "--ti=1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "no ROCm-capable device" in r.stderr          # no CPU fallback
    # --gpu-games default is derived: AVG_PRED_BATCH_SIZE / t * 2 (settings.h:163-171) = 32 / 4 * 2
    assert "Games per GPU 16, MCTS threads: 4, MCTS simulations 16" in r.stdout
    s = open(tmp_path / "log" / "settings.txt").read()
    assert "m(Mode [train/play])=learn" in s and "mcts(Number of MCTS simulations)=16" in s
    assert "cg(Number of games for comparison)=1000" in s


def test_unknown_flag_is_an_error(exe, tmp_path):
    r = subprocess.run([exe, "--nope=1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 2 and "does not exist" in r.stderr


@pytest.mark.gpu
def test_learn_mode_generates_reference_format_samples(exe, tmp_path):
    """one full AlphaZeroTrainer::train iteration in the C++ host: self-play -> azr_nn_train -> arena new vs old (its
    samples join the replay buffer) -> accept -> benchmark vs Random / Script, with the reference's files"""
    r = subprocess.run([exe, "-m", "learn", "--mcts=8", "--gpu-games=16", "--blocks=1", "--ti=1", "--tg=6", "--dtype=bf16",
                        "--bs=64", "-e", "2", "--cg=4", "--ct=0"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert "Generated" in r.stdout and "simulations/s" in r.stdout
    assert "EPOCH 1" in r.stdout and "Loss Policy / Value:" in r.stdout
    assert "New samples generated from compare games" in r.stdout and "Model improved" in r.stdout
    assert "Model benchmark games played: 10" in r.stdout and "Model benchmark games played: 100" in r.stdout
    for f in ("log/azr-improvement-log.txt", "log/azr-benchmark-log.txt", "log/azr-nn-training-log.txt",
              "checkpoints/best-checkpoint.bin", "checkpoints/checkpoint-iter-0.bin", "checkpoints/temp.bin"):
        assert os.path.getsize(tmp_path / f) > 0, f
    from test_log_grammar import check   # the reference's own grammar (tests/golden/ref_logs.json) and log_chart.py's parsing
    assert len(check("improvement", open(tmp_path / "log" / "azr-improvement-log.txt").read())) == 1
    assert len(check("benchmark", open(tmp_path / "log" / "azr-benchmark-log.txt").read())) == 1
    assert [len(r) for r in check("nn", open(tmp_path / "log" / "azr-nn-training-log.txt").read())] == [4]
    imp = open(tmp_path / "log" / "azr-improvement-log.txt").read().strip().split(",")
    assert imp[0] == "0" and len(imp) == 4
    nnl = open(tmp_path / "log" / "azr-nn-training-log.txt").read().strip().rstrip(",").split(",")
    assert len(nnl) == 4 and all(float(x) > 0 for x in nnl)          # 2 epochs x (policy, value)
    raw = open(tmp_path / "data" / "training_samples.bin", "rb").read()
    n = int(np.frombuffer(raw[:8], np.uint64)[0])
    assert n > 0 and len(raw) == 8 + n * 265                                  # alphazero_nn_data.cpp:123-130
    rec = np.frombuffer(raw[8:], np.uint8).reshape(n, 265)
    pi = rec[:, 93:].copy().view(np.float32).reshape(n, 43)
    z = rec[:, 89:93].copy().view(np.float32).reshape(n)
    assert np.allclose(pi.sum(1), 1, atol=1e-4) and set(np.unique(z)) <= {-1.0, 0.0, 1.0}
    assert os.path.exists(tmp_path / "checkpoints" / "latest-checkpoint.bin")  # missing checkpoint => init + save
    # exactly TRAIN_ITERATION_GAMES self-play games were played to their end (Counter::hasNext, alphazero_trainer.cpp:83)
    assert "Self-play: 6 games" in r.stdout


@pytest.mark.gpu
def test_play_mode_az_vs_az(exe, tmp_path):
    for extra in ([], ["--dtype=f16"]):   # (the default bf16 tower, and the same kernels on fp16 operands)
        r = subprocess.run([exe, "-m", "play", "--p1=az", "--p2=az", "--mcts=4", "--gpu-games=8", "--blocks=1", "--cg=8"] + extra,
                           cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        tail = r.stdout.strip().split("\n")[-4:]
        assert tail[0] == "Games: 8"
        d, p1, p2 = (int(t.split(":")[1]) for t in tail[1:])
        assert d + p1 + p2 == 8


@pytest.mark.gpu
def test_play_mode_config0_az_vs_script(exe, tmp_path):
    """BASELINE configs[0]: `-m play --mcts=16 --cg=N`, AlphaZero (p1, default) vs ScriptPlayer (p2, default)"""
    r = subprocess.run([exe, "-m", "play", "--mcts=16", "--cg=12", "--gpu-games=4", "--blocks=1"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    tail = r.stdout.strip().split("\n")[-4:]
    assert tail[0] == "Games: 12"
    d, p1, p2 = (int(t.split(":")[1]) for t in tail[1:])
    assert d + p1 + p2 == 12


@pytest.mark.gpu
def test_play_mode_script_vs_random(exe, tmp_path):
    r = subprocess.run([exe, "-m", "play", "--p1=sp", "--p2=rp", "--cg=40", "--gpu-games=8", "--blocks=1"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    tail = r.stdout.strip().split("\n")[-4:]
    d, p1, p2 = (int(t.split(":")[1]) for t in tail[1:])
    assert tail[0] == "Games: 40" and d + p1 + p2 == 40 and p1 > p2   # the scripted player beats the random one
