#!/usr/bin/env python3
"""Where every evaluation of the 20-block net stands against a float64 evaluation of the same graph (PyTorch on the CPU,
tests/torch_train_ref.py): the oracle's fp32 CPU net (stand-in for the reference's fp32 TensorFlow session), and the engine's
NET_F32 (fp32 VALU), NET_F32X (fp16-pair MFMA), NET_F16 and NET_BF16 towers, on the 128 distinct boards of tests/test_gpu_net.py.
    python tools/net_precision.py [blocks]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azr_testlib as T   # noqa: E402
import torch_train_ref as R   # noqa: E402


def f64_forward(flat, blocks, x):
    net = R.AzrNet(blocks, flat).double().eval()
    with torch.no_grad():
        lg, v = net(torch.from_numpy(R.planes_from_in88(x)).double())
        return torch.softmax(lg, 1).numpy(), v.numpy()


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    P = importlib.import_module("alphazero-risk_amd")
    g = np.unique(np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"], axis=0)
    x = g[np.linspace(0, len(g) - 1, 128).astype(int)].copy()
    flat = T.make_net_flat(blocks, seed=3, perturb_bn=True)
    rpi, rv = f64_forward(flat, blocks, x)
    orc = T.oracle()
    net = T.OrcNet(blocks, flat.ctypes.data_as(T.f32p))
    opi, ov = np.zeros((len(x), 43), np.float32), np.zeros(len(x), np.float32)
    orc.orc_net_forward_mt(C.byref(net), T.ptr(x), len(x), T.ptr(opi), T.ptr(ov), 16)
    rows = [("oracle fp32 (CPU)", opi, ov)]
    for name, dt in (("NET_F32  (VALU)", P.NET_F32), ("NET_F32X (fp16 pairs)", P.NET_F32X), ("NET_F16", P.NET_F16), ("NET_BF16", P.NET_BF16)):
        eng = P.Engine(len(x), blocks=blocks, sims=1, dtype=dt, node_capacity=64)
        eng.set_weights(flat)
        pi, v = eng.predict(x)
        eng.close()
        rows.append((name, pi, v))
    print(f"B = {blocks}, {len(x)} boards; max |d pi|, max |d v| against the float64 graph, and against the oracle")
    for name, pi, v in rows:
        print(f"  {name:24s} vs f64: {np.abs(pi - rpi).max():.2e} {np.abs(v - rv).max():.2e}    vs oracle: {np.abs(pi - opi).max():.2e} {np.abs(v - ov).max():.2e}")


if __name__ == "__main__":
    main()
