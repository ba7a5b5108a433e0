#!/usr/bin/env python3
"""Second half of the Winograd study (tools/winograd_study.py has the precision numbers): does a Winograd F(2x2, 3x3) tower with
16-bit operands pick the same moves as the fp32 search?  The 96 seeded golden roots of tests/test_gpu_net.py's move-agreement test
(B = 20, S = 100, T = 1, random-init weights) are searched by the ORACLE's search (oracle/azr_oracle.c) on
    its own fp32 CPU net                                  (the reference line)
    a direct conv with bf16 / fp16 operand rounding       (what the NET_BF16 / NET_F16 towers compute, emulated)
    a Winograd F(2x2, 3x3) conv with bf16 / fp16 V and U  (emulated: transforms in fp32, operands rounded, products / sums fp32)
Every root runs in its own thread; the threads' net evaluations meet in one batched PyTorch call per search step.
    python tools/winograd_moves.py [--roots 96] [--variants direct_bf16,direct_f16,wino_bf16,wino_f16]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azr_testlib as T   # noqa: E402
import torch_train_ref as R   # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G_ = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def rne_bf16(x):
    u = x.contiguous().view(torch.int32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & ~0xffff
    return u.view(torch.float32)


def rne_f16(x):
    return x.to(torch.float16).to(torch.float32)


def scale_pow2(w):
    m = float(w.abs().max())
    return 2.0 ** (13 - int(np.floor(np.log2(m)))) if m > 0 else 1.0


class EmuNet:
    """the net of tests/torch_train_ref.AzrNet in inference mode with the tower convs replaced by an emulation of 16-bit operands;
    per-layer operand tensors are prepared once"""

    def __init__(self, blocks, flat, mode, rnd, scaled):
        self.net = R.AzrNet(blocks, flat).float().eval()
        self.mode, self.rnd = mode, rnd
        self.prep = {}
        for b in range(blocks):
            for ab in "ab":
                w = self.net.p[f"b{b}{ab}_w"].detach()               # HWIO
                if mode == "direct":
                    s = scale_pow2(w) if scaled else 1.0
                    self.prep[id(self.net.p[f"b{b}{ab}_w"])] = (rnd(w * s) / s).permute(3, 2, 0, 1).contiguous()
                else:
                    U = G_ @ w.permute(3, 2, 0, 1) @ G_.T            # [co, ci, 4, 4]
                    s = scale_pow2(U) if scaled else 1.0
                    self.prep[id(self.net.p[f"b{b}{ab}_w"])] = (rnd(U * s) / s).permute(2, 3, 1, 0).reshape(16, 256, 256).contiguous()   # [pos][ci][co]
        conv0 = R.AzrNet._conv
        emu = self

        def conv(net_self, a, w):
            if w.shape[0] != 3 or a.shape[1] != 256:
                return conv0(net_self, a, w)
            p = emu.prep[id(w)]
            if emu.mode == "direct":
                return F.conv2d(emu.rnd(a), p, padding=1)
            n = a.shape[0]
            ap = torch.zeros((n, 256, 10, 8), dtype=torch.float32)
            ap[:, :, 1:8, 1:7] = a
            d = torch.stack([ap[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for ty in range(4) for tx in range(3)], 2)   # [n, ci, 12, 4, 4]
            V = emu.rnd(BT @ d @ BT.T)
            Vm = V.permute(3, 4, 0, 2, 1).reshape(16, n * 12, 256)    # [pos][n * tile][ci]
            M = torch.bmm(Vm, p).reshape(4, 4, n, 12, 256).permute(2, 4, 3, 0, 1)   # [n, co, 12, 4, 4]
            Y = AT @ M @ AT.T
            out = torch.zeros((n, 256, 8, 6), dtype=torch.float32)
            for ty in range(4):
                for tx in range(3):
                    out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[:, :, ty * 3 + tx]
            return out[:, :, :7, :]
        self.conv = conv

    def __call__(self, in88):
        old = R.AzrNet._conv
        R.AzrNet._conv = self.conv
        try:
            with torch.no_grad():
                lg, v = self.net(torch.from_numpy(R.planes_from_in88(in88)))
                return torch.softmax(lg, 1).numpy(), v.numpy()
        finally:
            R.AzrNet._conv = old


class Batcher:
    """net evaluations of the search threads, one batched call per step: a thread hands in its position and sleeps until the batch of
    all threads that are still searching has been evaluated"""

    def __init__(self, net, n_threads):
        self.net, self.alive = net, n_threads
        self.cv = threading.Condition()
        self.req, self.res, self.gen = [], {}, 0

    def evaluate(self, x88):
        with self.cv:
            my = len(self.req)
            self.req.append(x88)
            gen = self.gen
            if len(self.req) == self.alive:
                self._flush()
            else:
                while self.gen == gen:
                    self.cv.wait()
            return self.res[gen][0][my], self.res[gen][1][my]

    def leave(self):
        with self.cv:
            self.alive -= 1
            if self.alive > 0 and len(self.req) == self.alive:
                self._flush()

    def _flush(self):
        pi, v = self.net(np.stack(self.req))
        self.res = {self.gen: (pi, v)}
        self.req = []
        self.gen += 1
        self.cv.notify_all()


def search_all(orc, cfg, states, seeds, batcher):
    n_out = np.zeros((len(states), 43), np.int64)
    mv = np.zeros(len(states), np.uint8)
    live = np.zeros(len(states), bool)

    @T.EVAL_FN
    def ev(ctx, in88, pi, v):
        x = np.ctypeslib.as_array(in88, shape=(88,)).copy()
        p, vv = batcher.evaluate(x)
        C.memmove(pi, np.ascontiguousarray(p, np.float32).ctypes.data, 43 * 4)
        v[0] = float(vv)

    def one(i):
        try:
            s, r = T.OrcState(), T.OrcRng()
            orc.orc_state_unpack(C.byref(s), T.ptr(states[i]))
            r.x = int(seeds[i])
            if orc.orc_game_status(C.byref(s), C.byref(cfg)) != -1:
                return
            live[i] = True
            m = orc.orc_mcts_create(C.byref(cfg))
            assert orc.orc_mcts_simulate(m, C.byref(s), C.byref(r), ev, None) == 0
            n32 = np.zeros(43, np.uint32)
            orc.orc_mcts_root_stats(m, C.byref(s), T.ptr(n32), None, None, None)
            n_out[i] = n32
            pi = np.zeros(43, np.float32)
            orc.orc_mcts_policy(m, C.byref(s), T.ptr(pi))
            mv[i] = orc.orc_pick_highest(T.ptr(pi))
            orc.orc_mcts_destroy(m)
        finally:
            batcher.leave()

    th = [threading.Thread(target=one, args=(i,)) for i in range(len(states))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return n_out, mv, live


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--roots", type=int, default=96)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--variants", default="direct_bf16,direct_f16,wino_bf16,wino_f16")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "6")))
    blocks = 20
    orc = T.oracle()
    gold = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    states = gold["states"][::29][:a.roots]
    seeds = np.arange(500, 500 + len(states), dtype=np.uint32)
    flat = T.make_net_flat(blocks, seed=20260002)
    cfg = T.default_settings(mcts_simulations=a.sims, mcts_threads=1)

    class Fp32:   # the oracle's own fp32 CPU net, batched the same way (the reference line)
        def __init__(self):
            self.net = T.OrcNet(blocks, flat.ctypes.data_as(T.f32p))

        def __call__(self, in88):
            x = np.ascontiguousarray(in88, np.uint8)
            pi, v = np.zeros((len(x), 43), np.float32), np.zeros(len(x), np.float32)
            orc.orc_net_forward_mt(C.byref(self.net), T.ptr(x), len(x), T.ptr(pi), T.ptr(v), 8)
            return pi, v

    t0 = time.time()
    ref_n, ref_mv, live = search_all(orc, cfg, states, seeds, Batcher(Fp32(), len(states)))
    print(f"reference: the oracle's search on its fp32 net, {int(live.sum())} live roots of {len(states)}, {time.time() - t0:.0f} s", flush=True)
    top2 = np.sort(ref_n, axis=1)[:, -2:]
    clear = live & ((top2[:, 1] - top2[:, 0]) > 2)
    for name in a.variants.split(","):
        mode, el = name.split("_")
        net = EmuNet(blocks, flat, "direct" if mode == "direct" else "wino", rne_bf16 if el == "bf16" else rne_f16, el == "f16")
        t0 = time.time()
        n, mv, _ = search_all(orc, cfg, states, seeds, Batcher(net, len(states)))
        same = (mv == ref_mv)[live]
        ident = (n == ref_n).all(1)[live]
        tv = (np.abs(n - ref_n).sum(1) / (2.0 * a.sims))[live].mean()
        print(f"{name:12s} identical argmax-N {same.sum()}/{len(same)}; on the {int(clear.sum())} clear roots {int((mv == ref_mv)[clear].sum())}/{int(clear.sum())}; "
              f"identical visit vectors {ident.sum()}/{len(ident)}; mean TV distance {tv:.4f}   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
