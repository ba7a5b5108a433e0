"""the native optimiser step (azr_nn_train*, csrc/azr_train.hip) against the same graph in PyTorch fp32 on
the CPU (tests/torch_train_ref.py::AzrNet, itself checked against the oracle's forward pass in tests/test_train.py).
"parity unpinned": the reference's step is a TensorFlow session (absent here); what is pinned is build_graph.py's
arithmetic — losses, batch-statistics BN incl. the stem's axis-1 BN, gradients, L2, TF-formula Adam, moving averages —
and the reference's epoch loop (libstdc++ std::shuffle on minstd_rand0, remainder dropped, epoch-average losses).
Tolerances are stated per test; they cover fp32 summation-order differences only."""
import importlib
import os
import subprocess

import numpy as np
import pytest
import torch

import azr_testlib as T
from gpu_common import ROOT, pkg

pytestmark = pytest.mark.gpu
import torch_train_ref as train


def records(n, seed=0):
    """synthetic (s, pi, z) records on real encoded positions: pi random over a random support, z in {-1, 0, 1}"""
    g = np.load(os.path.join(T.GOLDEN, "encode.npz"))
    x = g["in88"]
    rng = np.random.default_rng(seed)
    x = x[rng.integers(0, len(x), n)]
    pi = rng.uniform(0.0, 1.0, (n, 43)).astype(np.float32) * (rng.uniform(0, 1, (n, 43)) < 0.3)
    pi[:, 42] += 1e-3
    pi = (pi / pi.sum(1, keepdims=True)).astype(np.float32)
    z = rng.integers(-1, 2, n).astype(np.float32)
    rec = np.zeros((n, 265), np.uint8)
    rec[:, 0] = x[:, 42]
    rec[:, 1:89] = x
    rec[:, 89:93] = z.view(np.uint8).reshape(n, 4)
    rec[:, 93:265] = pi.view(np.uint8).reshape(n, 172)
    return rec


def torch_step(blocks, flat, rec, margin=None):
    """loss, gradients (AZRW layout, without the L2 term) and post-step BN moving statistics from the PyTorch graph.
    margin (a list): receives the smallest non-zero |ReLU input| of the float64 forward — an fp32 forward whose rounding
    lands on the other side of such an input flips one mask, a discrete change of the gradients (measured: flips at
    2.5e-7 .. 6.4e-7, none at 1.1e-6, with either conv kernel)"""
    net = train.AzrNet(blocks, flat).double()
    net.train()
    in88, pi, z = train.unpack_records(rec)
    x = torch.from_numpy(train.planes_from_in88(in88)).double()
    relu0 = train.F.relu

    def relu(t, *a, **k):
        if margin is not None:
            v = t.detach().abs()
            v = v[v > 0]
            if v.numel():
                margin.append(float(v.min()))
        return relu0(t, *a, **k)

    train.F.relu = relu
    try:
        lp, lv, _ = net.losses(x, torch.from_numpy(pi).double(), torch.from_numpy(z).double())
    finally:
        train.F.relu = relu0
    (lp + lv).backward()
    g = np.zeros(net.count, np.float64)
    for name, off, shape in net.lay:
        n = int(np.prod(shape))
        if name.endswith("_bn"):
            key = name[:-3]
            c = shape[1]
            g[off:off + c] = net.p[key + "_g"].grad.numpy()
            g[off + c:off + 2 * c] = net.p[key + "_b"].grad.numpy()
        else:
            g[off:off + n] = net.p[name].grad.numpy().reshape(-1)
    return float(lp.detach()), float(lv.detach()), g, net.to_flat()


def kinds(blocks):
    k = np.zeros(train.layout(blocks)[1], np.uint8)
    for name, off, shape in train.layout(blocks)[0]:
        n = int(np.prod(shape))
        if name.endswith("_bn"):
            k[off:off + 2 * shape[1]] = 2
        elif name.endswith("_b"):
            k[off:off + n] = 2
        else:
            k[off:off + n] = 1
    return k


def tf_adam(w, g, m, v, t, k):
    """tf.train.AdamOptimizer update in fp32 on the trainable slots; kernels get the L2 gradient 2e-3 w"""
    w, g, m, v = (a.astype(np.float32).copy() for a in (w, g, m, v))
    g = np.where(k == 1, g + np.float32(2e-3) * w, g)
    lr_t = np.float32(1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t))
    tr = k > 0
    m2 = np.float32(0.9) * m + np.float32(1 - 0.9) * g
    v2 = np.float32(0.999) * v + np.float32(1 - 0.999) * g * g
    w2 = w - lr_t * m2 / (np.sqrt(v2) + np.float32(1e-8))
    return np.where(tr, w2, w), np.where(tr, m2, m), np.where(tr, v2, v)


@pytest.mark.parametrize("blocks,bs,gemm", [(1, 16, "split"), (2, 64, "split"), (2, 64, "r2"), (2, 64, "f32"), (1, 10, "split")])
def test_step_matches_torch_graph(blocks, bs, gemm, monkeypatch):
    """gemm = "split": the default step — t_conv_rs (2 boards per block in border-class row order) with the forward conv on fp16
    pairs (3 passes), the backward GEMMs on two bf16 parts, the normalise / statistics kernels fused into the convs' staging
    paths and epilogues; "r2": the same kernels as round 2 ran them (6-pass bf16 forward, separate normalise kernels:
    AZR_TRAIN_FWD=bf16, AZR_TRAIN_FUSE=0); "f32": the fp32-MFMA GEMMs
    (AZR_TRAIN_GEMM=f32); batch 10 (42 * 10 rows, not a multiple of the 32-deep k-tile) takes the fp32 kernels by itself.  Batches of up to 128 records run the small-batch
    conv kernel (t_conv_q) and 5-board weight-gradient slices."""
    for k in ("AZR_TRAIN_GEMM", "AZR_TRAIN_FWD", "AZR_TRAIN_FUSE", "AZR_TRAIN_FUSE_APPLY"):
        monkeypatch.delenv(k, raising=False)
    if gemm == "f32":
        monkeypatch.setenv("AZR_TRAIN_GEMM", gemm)
    if gemm == "r2":
        monkeypatch.setenv("AZR_TRAIN_FWD", "bf16")
        monkeypatch.setenv("AZR_TRAIN_FUSE", "0")
    P = pkg()
    flat = T.make_net_flat(blocks, seed=11, perturb_bn=True)
    # the first record seed whose float64 forward keeps every ReLU input at least 1e-6 away from zero: the gradient
    # tolerance below is then not at the mercy of one mask flipped by fp32 rounding (see torch_step)
    for seed in range(blocks + 1, blocks + 33):
        rec = records(bs, seed=seed)
        margin = []
        rlp, rlv, rg, rflat = torch_step(blocks, flat, rec, margin)
        if min(margin) >= 1e-6:
            break
    else:
        pytest.fail("no record seed with a ReLU margin of 1e-6")
    eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64, test_hooks=gemm in ("r2", "f32"))   # the switches exist in libazr_hip_test.so only
    eng.set_weights(flat)
    lp, lv = eng.train_batch(rec)
    g = eng.train_grads()
    w1 = eng.get_weights()
    # losses: fp32 vs float64 reference
    assert abs(lp - rlp) <= 2e-5 * max(1, abs(rlp)) and abs(lv - rlv) <= 2e-5, (lp, rlp, lv, rlv)
    # gradients per tensor: max error relative to the tensor's largest gradient
    k = kinds(blocks)
    for name, off, shape in train.layout(blocks)[0]:
        n = int(np.prod(shape)) if not name.endswith("_bn") else 2 * shape[1]
        a, b = g[off:off + n].astype(np.float64), rg[off:off + n]
        scale = max(np.abs(b).max(), 1e-6)
        err = np.abs(a - b).max() / scale
        assert err <= 2e-3, (name, err, scale)
    # BN moving statistics after the step (momentum 0.99, unbiased variance)
    mov = (k == 0)
    assert np.allclose(w1[mov], rflat[mov], rtol=2e-5, atol=1e-6)
    # Adam on the engine's own gradients reproduces the engine's weights (first step: m = v = 0)
    w_ref, _, _ = tf_adam(flat, g, np.zeros_like(flat), np.zeros_like(flat), 1, k)
    tr = k > 0
    assert np.abs(w1[tr] - w_ref[tr]).max() <= 2e-7, np.abs(w1[tr] - w_ref[tr]).max()
    # inference after training uses the updated weights (refold / repack happened)
    x = rec[:8, 1:89].copy()
    pi, v = eng.predict(x)
    eng2 = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    eng2.set_weights(w1)
    pi2, v2 = eng2.predict(x)
    assert (pi.view(np.uint32) == pi2.view(np.uint32)).all() and (v == v2).all()
    eng.close(); eng2.close()


def test_fused_normalise_kernels_change_no_bit(monkeypatch):
    """the normalise steps computed inside the consuming convs' staging paths (t_conv_rs<.., PRO>; default) against the same
    arithmetic as kernels of their own (AZR_TRAIN_FUSE_APPLY=0): same formula, same inputs, same statistics -> the weights after
    three optimiser steps are equal bit for bit."""
    P = pkg()
    blocks = 2
    flat = T.make_net_flat(blocks, seed=5, perturb_bn=True)
    out = {}
    monkeypatch.setenv("AZR_TRAIN_CONVQ", "0")   # (the small-batch conv kernel exists in fused form only: compare like with like)
    for bs in (64, 48):
        rec = records(3 * bs, seed=17)
        for mode in ("1", "0"):
            monkeypatch.setenv("AZR_TRAIN_FUSE_APPLY", mode)
            eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64, test_hooks=True)   # (both sides: AZR_TRAIN_CONVQ=0 is a hook too)
            eng.set_weights(flat)
            losses = [eng.train_batch(rec[t * bs:(t + 1) * bs]) for t in range(3)]
            out[mode] = (eng.get_weights(), eng.train_grads(), losses)
            eng.close()
        assert out["1"][2] == out["0"][2], (bs, out["1"][2], out["0"][2])
        assert (out["1"][1].view(np.uint32) == out["0"][1].view(np.uint32)).all(), bs
        assert (out["1"][0].view(np.uint32) == out["0"][0].view(np.uint32)).all(), bs
        assert np.abs(out["1"][0] - flat).max() > 1e-3


@pytest.mark.parametrize("blocks,bs", [(1, 16), (2, 64)])
def test_step_on_unfiltered_records(blocks, bs):
    """the same step on the FIRST record seed (seed = blocks), whatever its ReLU margins are: element-wise parity is only
    promised away from ReLU ties (test above), so here each tensor is bounded by its relative L2 error plus the share of its
    entries beyond the element-wise tolerance — a regression on ordinary inputs stays visible even when a mask flips"""
    P = pkg()
    flat = T.make_net_flat(blocks, seed=11, perturb_bn=True)
    rec = records(bs, seed=blocks)
    margin = []
    rlp, rlv, rg, _ = torch_step(blocks, flat, rec, margin)
    eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    eng.set_weights(flat)
    lp, lv = eng.train_batch(rec)
    g = eng.train_grads().astype(np.float64)
    eng.close()
    assert abs(lp - rlp) <= 2e-5 * max(1, abs(rlp)) and abs(lv - rlv) <= 2e-5, (lp, rlp, lv, rlv)
    worst_l2, worst_share = 0.0, 0.0
    for name, off, shape in train.layout(blocks)[0]:
        n = int(np.prod(shape)) if not name.endswith("_bn") else 2 * shape[1]
        a, b = g[off:off + n], rg[off:off + n]
        nb = np.linalg.norm(b)
        if nb == 0:
            continue
        l2 = np.linalg.norm(a - b) / nb
        share = (np.abs(a - b) > 2e-3 * max(np.abs(b).max(), 1e-6)).mean()
        worst_l2, worst_share = max(worst_l2, l2), max(worst_share, share)
        assert l2 <= 2e-2 and share <= 2e-2, (name, l2, share, min(margin))
    print(f"unfiltered seed {blocks}: smallest |ReLU input| {min(margin):.1e}, worst relative L2 {worst_l2:.1e}, worst share beyond 2e-3 {worst_share:.1e}")


def test_conv_kernel_families_at_the_reference_batch(monkeypatch):
    """BATCH_SIZE 512 (the reference's; 256 two-board blocks of t_conv_rs, 15 slices of 7 five-board groups of t_wgrad_g5 — the
    shapes the learn loop runs): one step with the default kernels (fp16-pair forward, fused normalise / statistics), one as
    round 2 ran them (6-pass bf16 forward, separate kernels) and one with the fp32-MFMA GEMMs, against the float64 PyTorch
    graph.  With 27 M ReLU inputs per step some lie within fp32 rounding of zero (2.9e-8 here) and every implementation flips
    its own few masks, so single gradient entries differ by up to 6e-3 of the tensor's largest between ANY two of them; the
    test bounds the relative L2 error per tensor instead (measured: 3e-4 .. 1e-3 for the conv kernels of all families alike,
    1e-6 for the head tensors)"""
    P = pkg()
    blocks, bs = 2, 512
    flat = T.make_net_flat(blocks, seed=21, perturb_bn=True)
    rec = records(bs, seed=77)
    rlp, rlv, rg, _ = torch_step(blocks, flat, rec)
    for mode in ("default", "r2", "f32"):
        for k in ("AZR_TRAIN_GEMM", "AZR_TRAIN_FWD", "AZR_TRAIN_FUSE"):
            monkeypatch.delenv(k, raising=False)
        if mode == "r2":
            monkeypatch.setenv("AZR_TRAIN_FWD", "bf16")
            monkeypatch.setenv("AZR_TRAIN_FUSE", "0")
        if mode == "f32":
            monkeypatch.setenv("AZR_TRAIN_GEMM", "f32")
        eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64, test_hooks=mode != "default")
        eng.set_weights(flat)
        lp, lv = eng.train_batch(rec)
        g = eng.train_grads().astype(np.float64)
        eng.close()
        assert abs(lp - rlp) <= 2e-6 * max(1, abs(rlp)) and abs(lv - rlv) <= 2e-6, (mode, lp, rlp, lv, rlv)
        for name, off, shape in train.layout(blocks)[0]:
            if not name.endswith("_w"):
                continue
            n = int(np.prod(shape))
            err = np.linalg.norm(g[off:off + n] - rg[off:off + n]) / np.linalg.norm(rg[off:off + n])
            assert err <= (3e-3 if name[0] in "sb" else 1e-5), (mode, name, err)   # stem / block kernels | head tensors


@pytest.mark.parametrize("bs", [512, 64, 48, 16])
def test_weight_gradient_formulations_agree(monkeypatch, bs):
    """t_wgrad_g5 (the product: a k-step is one board row of five boards, the taps share fragments out of a ring of three rows, slices of
    whole groups) against t_wgrad_rs (rows in memory order, one fragment read per tap; libazr_hip_test.so, AZR_TRAIN_WGRAD=rs): the same
    products summed in another order — identical losses, conv-kernel gradients to fp32 summation noise.  48 and 16 records end in a
    partial group of boards (48 = 9 slices of 5 + 3, 16 = 3 x 5 + 1)."""
    P = pkg()
    blocks = 2
    flat = T.make_net_flat(blocks, seed=21, perturb_bn=True)
    rec = records(bs, seed=77)
    out = {}
    for mode in ("g5", "rs"):
        monkeypatch.delenv("AZR_TRAIN_WGRAD", raising=False)
        if mode == "rs":
            monkeypatch.setenv("AZR_TRAIN_WGRAD", "rs")
        eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64, test_hooks=mode == "rs")
        eng.set_weights(flat)
        out[mode] = (eng.train_batch(rec), eng.train_grads().astype(np.float64))
        eng.close()
    assert out["g5"][0] == out["rs"][0]
    a, b = out["g5"][1], out["rs"][1]
    for name, off, shape in train.layout(blocks)[0]:
        n = int(np.prod(shape)) if not name.endswith("_bn") else 2 * shape[1]
        x, y = a[off:off + n], b[off:off + n]
        if name.startswith("b") and name.endswith("_w"):
            err = np.linalg.norm(x - y) / np.linalg.norm(y)
            assert 0 < err <= 2e-6, (name, err)   # (0 would mean the switch did nothing)
        else:
            assert (x == y).all(), name           # everything else runs the same kernels on the same bits


def test_steps_are_reproducible_and_adam_state_persists():
    P = pkg()
    blocks, bs = 1, 32
    flat = T.make_net_flat(blocks, seed=3)
    rec = records(3 * bs, seed=9)
    k = kinds(blocks)
    outs = []
    for _ in range(2):
        eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_BF16, node_capacity=64)
        eng.set_weights(flat)
        w, m, v = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
        for t in range(3):
            eng.train_batch(rec[t * bs:(t + 1) * bs])
            g = eng.train_grads()
            mov = eng.get_weights()
            w, m, v = tf_adam(w, g, m, v, t + 1, k)
            w[k == 0] = mov[k == 0]
            assert np.abs(mov[k > 0] - w[k > 0]).max() <= 5e-7
        # a different batch size rebuilds the activation buffers but keeps the Adam moments and step count
        eng.train_batch(rec[:bs // 2])
        g = eng.train_grads()
        mov = eng.get_weights()
        w, m, v = tf_adam(w, g, m, v, 4, k)
        assert np.abs(mov[k > 0] - w[k > 0]).max() <= 5e-7
        outs.append(eng.get_weights())
        eng.close()
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all()   # atomic-free reductions: bit-reproducible


def test_epoch_loop_follows_reference_shuffle(tmp_path):
    """azr_nn_train == the reference's loop: std::shuffle(minstd_rand0) per epoch, floor(n / bs) steps, remainder
    dropped, epoch-average losses; checked against train_batch driven by libstdc++'s own std::shuffle"""
    exe = str(tmp_path / "shuffle_probe")
    subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "helpers", "shuffle_probe.cpp")])
    P = pkg()
    blocks, bs, n, epochs, state = 1, 16, 70, 2, 20260001
    out = subprocess.check_output([exe, str(n), str(state), str(epochs)]).decode().split("\n")
    perms = [np.array(out[e].split(), int) for e in range(epochs)]
    end_state = int(out[epochs])
    flat = T.make_net_flat(blocks, seed=4)
    rec = records(n, seed=5)
    a = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    a.set_weights(flat)
    hist, st = a.train(rec, epochs, batch_size=bs, rng_state=state)
    assert st == end_state
    b = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    b.set_weights(flat)
    for e in range(epochs):
        lp = lv = np.float32(0)
        for c in range(n // bs):
            l = b.train_batch(rec[perms[e][c * bs:(c + 1) * bs]])
            lp += np.float32(l[0]); lv += np.float32(l[1])
        assert hist[e] == (float(lp / np.float32(n // bs)), float(lv / np.float32(n // bs)))
    assert (a.get_weights().view(np.uint32) == b.get_weights().view(np.uint32)).all()
    # fewer records than a batch: no step, NaN losses, weights untouched
    w = a.get_weights()
    hist, _ = a.train(rec[:bs - 1], 1, batch_size=bs, rng_state=1)
    assert np.isnan(hist[0][0]) and (a.get_weights() == w).all()
    a.close(); b.close()


def test_training_reduces_loss_on_fixed_batch():
    P = pkg()
    blocks, bs = 2, 64
    eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_BF16, node_capacity=64)
    eng.init_random(5)
    rec = records(bs, seed=1)
    first = eng.train_batch(rec)
    for _ in range(30):
        last = eng.train_batch(rec)
    # the policy targets are high-entropy random distributions, so CE has a floor near their entropy
    assert last[0] < first[0] - 0.3 and last[1] < 0.6 * first[1], (first, last)
    eng.close()


def test_weight_outside_the_fp16_pair_range_fails_loudly():
    """the training forward conv runs on fp16 pairs of 2^10 w; a conv weight with |w| >= 64 would become inf and the step's
    NaNs would spread silently.  The step detects it (and a loss that is not a number): azr_nn_train returns AZR_E_INVALID_ARGUMENT,
    the handle keeps the weights it had before the call, inference still works, and a later call with sane weights trains again."""
    P = pkg()
    blocks, bs = 1, 32
    flat = T.make_net_flat(blocks, seed=3)
    rec = records(2 * bs, seed=9)
    eng = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    bad = flat.copy()
    bad[9 * 13 * 256 + 28 + 1000] = 70.0          # one tower conv weight beyond 64
    eng.set_weights(bad)
    with pytest.raises(P.AzrError) as ei:
        eng.train(rec, 1, batch_size=bs, rng_state=5)
    assert ei.value.code == 1 and "fp16-pair" in str(ei.value)
    assert (eng.get_weights().view(np.uint32) == bad.view(np.uint32)).all()   # nothing half-trained was left behind
    pi, v = eng.predict(rec[:4, 1:89].copy())
    assert np.isfinite(pi).all() and np.isfinite(v).all()
    eng.set_weights(flat)
    hist, _ = eng.train(rec, 1, batch_size=bs, rng_state=5)
    assert np.isfinite(hist[0][0]) and np.isfinite(hist[0][1])
    ref = P.Engine(8, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    ref.set_weights(flat)
    hist2, _ = ref.train(rec, 1, batch_size=bs, rng_state=5)
    assert hist == hist2 and (eng.get_weights().view(np.uint32) == ref.get_weights().view(np.uint32)).all()   # the optimiser state started afresh
    eng.close(); ref.close()
