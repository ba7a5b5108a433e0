#!/usr/bin/env python3
"""Design study (CPU, float64): would a Winograd F(2x2, 3x3) tower with 16-bit MFMA operands stay inside the tolerance?

The 3x3 SAME conv of a 7x6 board (python/src/build_graph.py:43-49) as 4 x 3 = 12 output tiles of 2x2: per tile the 4x4 input
patch d is transformed V = Bt d B, the kernel g once U = G g Gt, M = sum_ci U * V at each of the 16 positions, Y = At M A.
12 tiles x 16 products = 192 multiplies per channel pair against the 324 tap-equivalents the direct tower issues today
(304 valid taps): 0.59 x the MFMA work.  What it costs is precision: V sums up to 4 activations and U mixes taps with weights
1/2 and 1/4 BEFORE the 16-bit rounding, and At M A then subtracts products of similar size.  This script emulates exactly that
rounding — V and U rounded to bf16 or fp16 (fp16 kernels scaled per layer like fold16 does), products and sums exact (float64),
everything else of the net float64 — on the 128 distinct boards and weights of tests/test_gpu_net.py at 20 blocks, next to the
direct conv with the same operand rounding.  Prints max |d pi|, max |d v| against the float64 graph.
    python tools/winograd_study.py [blocks]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azr_testlib as T   # noqa: E402
import torch_train_ref as R   # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G_ = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rne_bf16(x):
    t = x.to(torch.float32).contiguous()
    u = t.view(torch.int32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & ~0xffff
    return u.view(torch.float32).to(torch.float64)


def rne_f16(x):
    return x.to(torch.float16).to(torch.float64)


def scale_pow2(w):
    """fold16's per-layer scale: 2^k with max |2^k w| in [2^13, 2^14)"""
    m = float(w.abs().max())
    return 2.0 ** (13 - int(np.floor(np.log2(m)))) if m > 0 else 1.0


def winograd_conv(a, w, rnd, scaled):
    """a [n, ci, 7, 6], w HWIO [3, 3, ci, co] -> [n, co, 7, 6]; V and U rounded with `rnd`"""
    n, ci = a.shape[0], a.shape[1]
    co = w.shape[3]
    ap = torch.zeros((n, ci, 10, 8), dtype=torch.float64)
    ap[:, :, 1:8, 1:7] = a
    d = torch.stack([ap[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for ty in range(4) for tx in range(3)], 2)   # [n, ci, 12, 4, 4]
    V = rnd(BT @ d @ BT.T)
    g = w.permute(3, 2, 0, 1)                                     # [co, ci, 3, 3]
    U = G_ @ g @ G_.T                                             # [co, ci, 4, 4]
    s = scale_pow2(U) if scaled else 1.0
    U = rnd(U * s) / s
    M = torch.einsum("ocij,nctij->notij", U, V)
    Y = AT @ M @ AT.T                                             # [n, co, 12, 2, 2]
    out = torch.zeros((n, co, 8, 6), dtype=torch.float64)
    for ty in range(4):
        for tx in range(3):
            out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[:, :, ty * 3 + tx]
    return out[:, :, :7, :]


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "6")))
    flat = T.make_net_flat(blocks, seed=3, perturb_bn=True)
    net = R.AzrNet(blocks, flat).double().eval()
    g = np.unique(np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"], axis=0)
    in88 = g[np.linspace(0, len(g) - 1, 128).astype(int)]
    x = torch.from_numpy(R.planes_from_in88(in88)).double()
    conv0 = R.AzrNet._conv
    rows = []
    with torch.no_grad():
        lg, v = net(x)
        pi = torch.softmax(lg, 1)
        # self-check of the transform: exact operands must reproduce the direct conv
        a = torch.randn(2, 256, 7, 6, dtype=torch.float64)
        w = torch.randn(3, 3, 256, 256, dtype=torch.float64)
        err = float((winograd_conv(a, w, lambda t: t, False) - conv0(net, a, w)).abs().max())
        assert err < 1e-9, err
        for name, mode, rnd, scaled in [("direct conv, bf16 operands", "direct", rne_bf16, False),
                                        ("direct conv, fp16 operands (scaled kernels)", "direct", rne_f16, True),
                                        ("Winograd F(2x2,3x3), bf16 V and U", "wino", rne_bf16, False),
                                        ("Winograd F(2x2,3x3), fp16 V and U (scaled U)", "wino", rne_f16, True)]:
            def conv(self, a, w, mode=mode, rnd=rnd, scaled=scaled):
                if w.shape[0] != 3 or a.shape[1] != 256:
                    return conv0(self, a, w)       # stem and 1x1 heads stay exact in every variant
                if mode == "direct":
                    s = scale_pow2(w) if scaled else 1.0
                    return conv0(self, rnd(a), rnd(w * s) / s)
                return winograd_conv(a, w, rnd, scaled)
            R.AzrNet._conv = conv
            lg2, v2 = net(x)
            pi2 = torch.softmax(lg2, 1)
            dpi, dv = float((pi2 - pi).abs().max()), float((v2 - v).abs().max())
            flips = int((pi2.argmax(1) != pi.argmax(1)).sum())
            rows.append((name, dpi, dv, flips))
            print(f"{name:48s} max|dpi| {dpi:.2e}   max|dv| {dv:.2e}   argmax-pi flips {flips}/{len(x)}", flush=True)
    R.AzrNet._conv = conv0
    return rows


if __name__ == "__main__":
    main()
