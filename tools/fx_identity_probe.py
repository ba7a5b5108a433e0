#!/usr/bin/env python3
"""NET_F32X diagnostics: a 1-block net whose two tower convs are w0 x identity on the centre tap, so every conv output is a
single product a * w0 — any error beyond fp32 rounding is a data-path or representation error of the fp16-pair scheme."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azr_testlib as T   # noqa: E402

P = importlib.import_module("alphazero-risk_amd")
g = np.unique(np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"], axis=0)
x = g[np.linspace(0, len(g) - 1, 64).astype(int)].copy()
orc = T.oracle()
for w0, tap in ((1.0, 4), (0.3, 4), (0.3, 0), (0.7123, 8), (1.0, 1)):
    flat = T.make_net_flat(1, seed=3, perturb_bn=True)
    for name, off, n in T.net_layout(1):
        if name in ("b0a_w", "b0b_w"):
            w = np.zeros((9, 256, 256), np.float32)
            w[tap, np.arange(256), np.arange(256)] = w0
            flat[off:off + n] = w.reshape(-1)
    net = T.OrcNet(1, flat.ctypes.data_as(T.f32p))
    opi, ov = np.zeros((len(x), 43), np.float32), np.zeros(len(x), np.float32)
    orc.orc_net_forward_mt(C.byref(net), T.ptr(x), len(x), T.ptr(opi), T.ptr(ov), 8)
    out = []
    for dt in (P.NET_F32, P.NET_F32X):
        eng = P.Engine(len(x), blocks=1, sims=1, dtype=dt, node_capacity=64)
        eng.set_weights(flat)
        pi, v = eng.predict(x)
        eng.close()
        out.append((np.abs(pi - opi).max(), np.abs(v - ov).max()))
    print(f"w0 = {w0}, tap {tap}:  NET_F32 vs oracle {out[0][0]:.2e} {out[0][1]:.2e}   NET_F32X vs oracle {out[1][0]:.2e} {out[1][1]:.2e}")
