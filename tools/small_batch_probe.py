#!/usr/bin/env python3
"""Time azr_nn_predict (one bf16 net forward per call) at small batches: the split-channel tower k_tower_sc (default for <= 256 boards)
against one board per workgroup (k_tower_bf16<1>, AZR_TOWER_SC=0).
    python tools/small_batch_probe.py [n ...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]


def main():
    ns = [int(x) for x in sys.argv[1:]] or [1, 16, 64, 100, 128, 200, 256]
    eng = P.Engine(256, blocks=20, sims=1, dtype=P.NET_BF16, node_capacity=64, test_hooks=not os.environ.get("AZR_EXP_LIB"))   # AZR_TOWER_SC is a test hook
    eng.init_random(1)
    rng = np.random.default_rng(1)
    x = np.zeros((256, 88), np.uint8)
    x[:, :42] = rng.integers(1, 33, (256, 42)) | (rng.integers(0, 3, (256, 42)) << 6)
    x[:, 48:88] = rng.random((256, 10)).astype(np.float32).view(np.uint8)
    for n in ns:
        xx = np.ascontiguousarray(x[:n])
        for _ in range(5):
            eng.predict(xx)
        t0 = time.perf_counter()
        for _ in range(50):
            eng.predict(xx)
        print(f"n {n:4d}: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms per forward (AZR_TOWER_SC={os.environ.get('AZR_TOWER_SC', '1')})", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
