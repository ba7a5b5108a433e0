#!/usr/bin/env python3
"""Time the native optimiser step (azr_nn_train) at the reference's training shape: BATCH_SIZE 512, B residual blocks.
    python tools/train_bench.py [--blocks 20] [--bs 512] [--batches 8] [--epochs 2]
Prints ms per minibatch step and the achieved fp32 FLOP rate (3 x forward FLOPs of the dense-padded GEMMs)."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--bs", type=int, default=512)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--torch", action="store_true", help="also time the PyTorch-ROCm autograd implementation (train.py)")
    ap.add_argument("--dp-world", type=int, default=0,
                    help="time rank 0's share of a data-parallel step over this many ranks on ONE GPU: azr_nn_train_dp with an all-reduce "
                         "callback that returns at once (the sums stay local, so the losses mean nothing; every kernel and every stream "
                         "hand-over of the real step runs)")
    ap.add_argument("--native", action="store_true",
                    help="with --dp-world: the engine's own RCCL communicator (one rank, AZR_DP_LOOPBACK=1) instead of the callback: the all-reduces "
                         "are in-stream ncclAllReduce calls, as on a real multi-GPU run")
    a = ap.parse_args()
    if a.native:
        os.environ["AZR_DP_LOOPBACK"] = "1"
    if a.torch:   # torch bundles its own HIP runtime: let it initialise the device before the C-ABI library does
        import torch
        torch.cuda.init()
    eng = P.Engine(64, blocks=a.blocks, sims=8, dtype=P.NET_BF16, test_hooks=bool(a.native))   # AZR_DP_LOOPBACK is a test hook (libazr_hip_test.so)
    eng.init_random(1)
    # records from real self-play positions
    eng.selfplay_start(7)
    recs = []
    while sum(len(r) for r in recs) < a.bs * a.batches:
        eng.selfplay_run(64)
        recs.append(eng.drain())
    rec = np.concatenate(recs)[:a.bs * a.batches]
    calls = [0]
    if a.dp_world and a.native:
        # AZR_DP_LOOPBACK=1 (set before the first call): a one-rank communicator stands in for dp_world ranks — every collective of rank
        # 0's share runs in the stream, the sums stay local
        eng.dp_init(0, 1, P.dp_unique_id())

        def run(r, e):
            return eng.train_dp(r, e, None, 0, a.dp_world, batch_size=a.bs, rng_state=1)
    elif a.dp_world:
        def noop(ptr, count, dtype):
            calls[0] += 1

        def run(r, e):
            return eng.train_dp(r, e, noop, 0, a.dp_world, batch_size=a.bs, rng_state=1)
    else:
        def run(r, e):
            return eng.train(r, e, batch_size=a.bs, rng_state=1)
    run(rec[:a.bs], 1)   # allocate + warm up
    calls[0] = 0
    t0 = time.time()
    hist, _ = run(rec, a.epochs)
    dt = time.time() - t0
    steps = a.epochs * a.batches
    M = a.bs * 42
    flop = 3 * 2.0 * M * 256 * (9 * 16 + 2 * a.blocks * 9 * 256)
    if a.dp_world:
        print(f"blocks={a.blocks} bs={a.bs} rank 0 of {a.dp_world} ({a.bs // a.dp_world} records per rank): {1e3 * dt / steps:.2f} ms/step incl. "
              + (f"{calls[0] // steps} all-reduce hand-overs per step (callback returns at once)" if not a.native else
                 "every all-reduce an in-stream ncclAllReduce on a one-rank communicator"))
        return
    print(f"blocks={a.blocks} bs={a.bs}: {1e3 * dt / steps:.2f} ms/step, {flop * steps / dt / 1e12:.1f} TFLOP/s fp32 (dense-padded GEMM work), "
          f"losses {hist}")
    if a.torch:
        print(f"PyTorch-ROCm autograd (MIOpen fp32), same graph and batch: {1e3 * torch_time(a, eng, rec):.2f} ms/step")


def torch_time(a, eng, rec):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_train_ref as T
    tr = T.Trainer(a.blocks, eng.get_weights(), device="cuda:0", batch_size=a.bs, seed=0)
    tr.train(rec[:a.bs], 1)
    torch.cuda.synchronize()
    t0 = time.time()
    tr.train(rec, a.epochs)
    torch.cuda.synchronize()
    return (time.time() - t0) / (a.epochs * a.batches)


if __name__ == "__main__":
    main()
