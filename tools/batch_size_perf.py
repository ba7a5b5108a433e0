#!/usr/bin/env python3
"""Writes log/batch-size-perf.txt — the reference's one published performance artefact (python/log/batch-size-perf.txt, plotted
unchanged by python/src/log_chart.py:87-110 build_NN_batch_speed_chart) — for this engine.

The reference's LOG_PERFORMANCE block (neural_network/alphazero_gpu_cluster.cpp:54-65) times
`nn->processBatchPrediction()` — host tensor in, session run, host tensor out — over 100 batches after 10 warm-up batches and
prints total processing ns / total samples; its file holds one "batch, ns_per_sample" row per batch size 1 .. 1024.  The same
here through the same seam: azr_nn_predict (host in88 -> pi, v on the host, one call = one batch), 10 warm-up calls, 100 timed.

    python tools/batch_size_perf.py [--blocks 20] [--dtype bf16|f32x|f32] [--out log/batch-size-perf.txt]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BATCHES = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
# python/log/batch-size-perf.txt of the reference (unstated CUDA GPU, TensorFlow C++), ns per sample
REFERENCE_NS = {1: 8037884, 2: 4478465, 4: 2682581, 8: 2026081, 16: 1267505, 32: 674457, 64: 590644, 128: 473598, 256: 438234,
                512: 367633, 1024: 383206}


def measure(P, blocks, dtype, batches=BATCHES, warm=10, timed=100, inputs=None):
    eng = P.Engine(max(batches), blocks=blocks, sims=1, dtype=dtype, node_capacity=64)
    eng.init_random(20260002)
    rng = np.random.default_rng(1)
    if inputs is None:
        inputs = np.zeros((max(batches), 88), np.uint8)
        inputs[:, :42] = rng.integers(1, 33, (max(batches), 42)) | (rng.integers(0, 3, (max(batches), 42)) << 6)
        inputs[:, 48:88] = rng.random((max(batches), 10)).astype(np.float32).view(np.uint8)
    rows = []
    for b in batches:
        x = np.ascontiguousarray(inputs[:b])
        for _ in range(warm):
            eng.predict(x)
        t0 = time.perf_counter_ns()
        for _ in range(timed):
            eng.predict(x)
        ns = time.perf_counter_ns() - t0
        rows.append((b, ns // (timed * b)))
    eng.close()
    return rows


def write_log(path, rows):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "w") as f:
        for b, ns in rows:
            f.write(f"{b}, {ns}\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32x", "f32"])
    ap.add_argument("--out", default="log/batch-size-perf.txt")
    a = ap.parse_args()
    P = importlib.import_module("alphazero-risk_amd")
    dt = {"bf16": P.NET_BF16, "f16": P.NET_F16, "f32x": P.NET_F32X, "f32": P.NET_F32}[a.dtype]
    rows = measure(P, a.blocks, dt)
    write_log(a.out, rows)
    print(json.dumps({"file": a.out, "blocks": a.blocks, "dtype": a.dtype, "unit": "ns per sample (azr_nn_predict wall time / batch)",
                      "rows": [{"batch": b, "ns_per_sample": ns, "reference_ns_per_sample": REFERENCE_NS[b], "speedup": REFERENCE_NS[b] / max(ns, 1)}
                               for b, ns in rows]}))


if __name__ == "__main__":
    main()
