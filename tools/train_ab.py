#!/usr/bin/env python3
"""A/B of two builds of the optimiser step (timing experiments): one step at batch 512 / B = 2 on fixed records -> gradients to an .npz
(compare two runs bit for bit), then the 20-block step timed.   AZR_EXP_LIB=<lib.so> python tools/train_ab.py OUT.npz"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]
import azr_testlib as T   # noqa: E402


def records(n, seed):
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


out = sys.argv[1]
res = {}
for bs in (512, 64):
    eng = P.Engine(8, blocks=2, sims=1, dtype=P.NET_F32, node_capacity=64)
    eng.set_weights(T.make_net_flat(2, seed=21, perturb_bn=True))
    l = eng.train_batch(records(bs, 77))
    res[f"g{bs}"] = eng.train_grads()
    res[f"l{bs}"] = np.array(l)
    eng.close()
np.savez(out, **res)
eng = P.Engine(8, blocks=20, sims=1, dtype=P.NET_BF16, node_capacity=64)
eng.init_random(1)
rec = records(512 * 8, 5)
eng.train(rec[:512], 1, batch_size=512, rng_state=1)
t0 = time.time()
eng.train(rec, 3, batch_size=512, rng_state=1)
print(f"{os.environ.get('AZR_EXP_LIB', 'product')}: {1e3 * (time.time() - t0) / 24:.3f} ms/step at batch 512, B = 20", flush=True)
eng.close()
