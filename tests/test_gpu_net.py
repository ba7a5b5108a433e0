"""GPU parity, net rows a13-a14: HIP policy/value net against the oracle's fp32 restatement of
python/src/build_graph.py.  "parity unpinned" (no TensorFlow here, the reference pins no numeric output):
tolerances are stated per test."""
import ctypes as C
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

pytestmark = pytest.mark.gpu


def sample_inputs(n):
    g = np.load(os.path.join(T.GOLDEN, "encode.npz"))
    x = g["in88"]
    idx = np.linspace(0, len(x) - 1, n).astype(int)
    return x[idx].copy()


def oracle_forward(orc, flat, blocks, x):
    net = T.OrcNet(blocks, flat.ctypes.data_as(T.f32p))
    pi = np.zeros((len(x), 43), np.float32)
    v = np.zeros(len(x), np.float32)
    orc.orc_net_forward_mt(C.byref(net), T.ptr(x), len(x), T.ptr(pi), T.ptr(v), 8)
    return pi, v


@pytest.mark.parametrize("blocks", [1, 3])
def test_fp32_net_matches_oracle(orc, blocks):
    """fp32 VALU path: same math, different summation order -> |dpi| <= 2e-5, |dv| <= 2e-5"""
    P = pkg()
    x = sample_inputs(64)
    flat = T.make_net_flat(blocks, seed=3, perturb_bn=True)
    eng = P.Engine(64, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
    eng.set_weights(flat)
    assert (eng.get_weights() == flat).all()
    pi, v = eng.predict(x)
    rpi, rv = oracle_forward(orc, flat, blocks, x)
    assert np.abs(pi - rpi).max() <= 2e-5, np.abs(pi - rpi).max()
    assert np.abs(v - rv).max() <= 2e-5, np.abs(v - rv).max()
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5)
    # batch invariance: one-by-one == batched, bit for bit (the search relies on it)
    p1, v1 = eng.predict(x[:5])
    for i in range(5):
        pa, va = eng.predict(x[i:i + 1])
        assert (pa[0].view(np.uint32) == p1[i].view(np.uint32)).all() and va[0] == v1[i]
    eng.close()


def test_checkpoint_roundtrip(tmp_path):
    P = pkg()
    eng = P.Engine(4, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    eng.init_random(5)
    w = eng.get_weights()
    path = str(tmp_path / "ckpt.bin")
    eng.save(path)
    eng.init_random(6)
    assert not (eng.get_weights() == w).all()
    eng.load(path)
    assert (eng.get_weights() == w).all()
    eng.close()
