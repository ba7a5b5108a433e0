#!/usr/bin/env python3
"""Design study for the NET_F32X tower (CPU, float64): how far is a split-bf16 evaluation of the 20-block net from the exact one?

Every 3x3 conv's operands are replaced by sums of bf16 terms (round-to-nearest-even splits) and selected cross products are
kept; everything else (BN fold, ReLU, heads) stays float64, so the numbers isolate the error of the split itself.
    terms (a, w) / products                          what a kernel would issue
    2,2 / hh hl lh            3 MFMA passes
    2,2 / hh hl lh ll         4 passes
    3,2 / ...                 activations with a third term
    fp16 pairs                hi = rne16(x), lo = rne16((x - hi) * 2^11): 3 passes, cross terms scaled by 2^-11
Prints max |d pi|, max |d v| over the golden boards."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azr_testlib as T   # noqa: E402
import torch_train_ref as R   # noqa: E402


def bf(x):   # RNE to bf16, kept in float64
    t = x.to(torch.float32).contiguous()
    u = t.view(torch.int32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & ~0xffff
    return u.view(torch.float32).to(torch.float64)


def split16(x):
    """fp16 pair: hi = rne16(x), lo = rne16((x - hi) * 2^11) kept pre-scaled (a plain fp16 lo would be subnormal for |x| < 0.1)"""
    hi = x.to(torch.float16).to(torch.float64)
    lo = ((x - hi) * 2048.0).to(torch.float16).to(torch.float64)
    return [hi, lo]


def split(x, terms):
    out, r = [], x.clone()
    for _ in range(terms):
        h = bf(r)
        out.append(h)
        r = r - h
    return out


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    flat = T.make_net_flat(blocks, seed=3, perturb_bn=True)   # the weights and the 128 distinct boards of tests/test_gpu_net.py
    net = R.AzrNet(blocks, flat).double().eval()
    g = np.unique(np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"], axis=0)
    in88 = g[np.linspace(0, len(g) - 1, 128).astype(int)]
    x = torch.from_numpy(R.planes_from_in88(in88)).double()
    conv0 = R.AzrNet._conv
    with torch.no_grad():
        lg, v = net(x)
        pi = torch.softmax(lg, 1)
        for name, ta, tw, prods in [("bf16 (1 pass)", 1, 1, [(0, 0)]),
                                    ("2x2 terms, 3 passes hh hl lh", 2, 2, [(0, 0), (0, 1), (1, 0)]),
                                    ("2x2 terms, 4 passes", 2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
                                    ("3x2 terms, 5 passes", 3, 2, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]),
                                    ("3x3 terms, 6 passes", 3, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (0, 2)]),
                                    ("fp16 pairs, 3 passes hh + 2^-11 (hl + lh)", -1, -1, [(0, 0), (0, 1), (1, 0)]),
                                    ("fp32 operands (exact products)", 0, 0, None)]:
            def conv(self, a, w, ta=ta, tw=tw, prods=prods):
                if w.shape[0] != 3 or a.shape[1] != 256:
                    return conv0(self, a, w)       # stem and 1x1 heads: fp32 in every variant
                if prods is None:
                    return conv0(self, a.float().double(), w)
                if ta < 0:
                    sa, sw = split16(a), split16(w)
                    return conv0(self, sa[0], sw[0]) + (conv0(self, sa[0], sw[1]) + conv0(self, sa[1], sw[0])) / 2048.0
                sa, sw = split(a, ta), split(w, tw)
                out = 0
                for i, j in prods:
                    out = out + conv0(self, sa[i], sw[j])
                return out
            R.AzrNet._conv = conv
            lg2, v2 = net(x)
            pi2 = torch.softmax(lg2, 1)
            print(f"{name:38s} max|dpi| {float((pi2 - pi).abs().max()):.2e}   max|dv| {float((v2 - v).abs().max()):.2e}")
    R.AzrNet._conv = conv0


if __name__ == "__main__":
    main()
