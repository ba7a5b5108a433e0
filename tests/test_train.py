"""SURVEY §8 f-2 (provisional train step, tests/torch_train_ref.py): the PyTorch graph bound to the AZRW vector
reproduces the oracle's forward pass; losses, L2 term, Adam step and BN running statistics follow build_graph.py.
"parity unpinned" (TensorFlow absent): tolerances stated per test.  CPU."""
import ctypes as C
import importlib
import os

import numpy as np
import torch

import azr_testlib as T
from gpu_common import ROOT  # noqa: F401  (puts the repo root on sys.path)

import torch_train_ref as train


def inputs(n):
    g = np.load(os.path.join(T.GOLDEN, "encode.npz"))
    x = g["in88"]
    return x[np.linspace(0, len(x) - 1, n).astype(int)].copy()


def test_layout_matches_param_count(orc):
    for b in (1, 5, 20):
        assert train.layout(b)[1] == orc.orc_net_param_count(b)


def test_planes_match_oracle(orc):
    x = inputs(40)
    p = train.planes_from_in88(x)                      # [n,13,7,6]
    t = np.zeros((42, 13), np.float32)
    for i in range(len(x)):
        orc.orc_planes(T.ptr(x[i]), T.ptr(t))
        assert np.array_equal(p[i].reshape(13, 42).T, t)


def test_eval_forward_matches_oracle(orc):
    """same fp32 math, different summation order: |dpi| <= 2e-5, |dv| <= 2e-5"""
    B = 2
    flat = T.make_net_flat(B, seed=5, perturb_bn=True)
    x = inputs(24)
    net = train.AzrNet(B, flat).eval()
    with torch.no_grad():
        logits, v = net(torch.from_numpy(train.planes_from_in88(x)))
        pi = torch.softmax(logits, 1).numpy()
    onet = T.OrcNet(B, flat.ctypes.data_as(T.f32p))
    rpi = np.zeros((len(x), 43), np.float32); rv = np.zeros(len(x), np.float32)
    orc.orc_net_forward(C.byref(onet), T.ptr(x), len(x), T.ptr(rpi), T.ptr(rv))
    assert np.abs(pi - rpi).max() <= 2e-5 and np.abs(v.numpy() - rv).max() <= 2e-5
    assert np.array_equal(net.to_flat(), flat)          # round trip of the flat vector


def _records(n, seed=0):
    rng = np.random.default_rng(seed)
    x = inputs(n)
    pi = rng.dirichlet(np.ones(43), n).astype(np.float32)
    z = rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)
    rec = np.zeros((n, 265), np.uint8)
    rec[:, 0] = x[:, 42]
    rec[:, 1:89] = x
    rec[:, 89:93] = z.view(np.uint8).reshape(n, 4)
    rec[:, 93:] = pi.view(np.uint8).reshape(n, 172)
    return rec, x, pi, z


def test_loss_terms_and_first_adam_step():
    B = 1
    flat = T.make_net_flat(B, seed=9)
    rec, x, pi, z = _records(64)
    tr = train.Trainer(B, flat, batch_size=64, seed=1)
    net = tr.net.train()
    xb = torch.from_numpy(train.planes_from_in88(x))
    lp, lv, l2 = net.losses(xb, torch.from_numpy(pi), torch.from_numpy(z))
    # L2 term = 1e-3 * sum of squares of the 2B+3 conv kernels and the 3 dense kernels (not biases, not BN)
    names = [n for n, _, _ in train.layout(B)[0] if n.endswith("_w")]
    assert len(names) == 2 * B + 3 + 3
    want = 1e-3 * sum(float((flat[o:o + int(np.prod(s))] ** 2).sum()) for n, o, s in train.layout(B)[0] if n.endswith("_w"))
    assert abs(float(l2) - want) <= 1e-4 * want
    assert 3.0 < float(lp) < 5.0            # ~ cross entropy of a near-uniform policy against Dirichlet targets
    # first Adam step from zero moments: every parameter with a non-zero gradient moves by lr * g / (|g| + eps)
    before = tr.flat()
    (lp + lv + l2).backward()
    g = {k: p.grad.clone() for k, p in net.p.items()}
    tr.opt.step()
    after = tr.flat()
    for name, off, shape in train.layout(B)[0]:
        if name.endswith("_bn"):
            continue
        n = int(np.prod(shape))
        gg = g[name].numpy().reshape(-1)
        step = after[off:off + n] - before[off:off + n]
        want_step = -1e-3 * gg / (np.abs(gg) + 1e-8)
        assert np.allclose(step, want_step, atol=2e-6), name


def test_training_reduces_loss_and_updates_bn_running_stats():
    B = 1
    flat = T.make_net_flat(B, seed=11)
    rec, *_ = _records(256, seed=3)
    tr = train.Trainer(B, flat, batch_size=64, seed=2)
    hist = tr.train(rec, epochs=6)
    assert len(hist) == 6 and hist[-1][0] < hist[0][0] and hist[-1][1] <= hist[0][1] + 1e-3
    new = tr.flat()
    for name, off, shape in train.layout(B)[0]:
        if name.endswith("_bn"):
            c = shape[1]
            mean_new, var_new = new[off + 2 * c:off + 3 * c], new[off + 3 * c:off + 4 * c]
            assert not np.allclose(mean_new, 0) or name == "v_bn"     # moving mean moved (momentum 0.99)
            assert (var_new > 0).all()
    # remainder records of an epoch are dropped: 100 records, batch 64 -> one minibatch per epoch
    assert len(tr.train(rec[:100], epochs=2)) == 2
    assert tr.train(rec[:10], epochs=2) == []
