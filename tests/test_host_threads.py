"""CPU: robustness of the host side's fan-outs.

* azrhost::forEachGpu (alphazero-risk_amd/host/azr_host.cpp) — one host thread per GPU, the reference's structure
  (alphazero_trainer.cpp:48-57, game.cpp:277-312): a failure inside a thread body reaches the caller as an exception after every
  thread has been joined (round-3 verdict: it used to end AlphaZero_Risk_hip in std::terminate).
* learn.Deadline — the watchdog of a multi-rank learn iteration: a rank that waits in a collective for a rank that is gone ends
  itself with exit code 124 instead of waiting for ever (round-3 advice)."""
import importlib
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "alphazero-risk_amd", "host")


def test_exception_in_a_per_gpu_thread_reaches_the_caller(tmp_path):
    sys.path.insert(0, ROOT)
    importlib.import_module("alphazero-risk_amd").build()
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = str(tmp_path / "probe")
    csrc = os.path.join(ROOT, "alphazero-risk_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "helpers", "host_threads_probe.cpp"),
                           os.path.join(HOST, "azr_host.o"), "-o", exe, "-L" + csrc, "-lazr_hip", "-Wl,-rpath," + csrc])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "caught: compare games on gpu 2: arena_run: engine said no" in r.stdout and r.stdout.strip().endswith("OK")


def test_deadline_ends_a_rank_whose_partner_is_gone():
    """two ranks over gloo on the CPU; rank 1 leaves; rank 0 arms learn.Deadline(2 s) and enters an all_reduce that can never
    complete: it must end with exit code 124 and the watchdog's message, not hang"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(ROOT, "tests", "helpers", "deadline_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker, os.path.join(ROOT, "alphazero-risk_amd", "learn.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    _, err0 = procs[0].communicate(timeout=120)
    procs[1].wait(timeout=60)
    assert procs[0].returncode == 124, (procs[0].returncode, err0[-1000:])
    assert "rank 0 spent more than 2 s in 'training and weight hand-over': a rank is missing from a collective" in err0
