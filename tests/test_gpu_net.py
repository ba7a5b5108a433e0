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


@pytest.mark.parametrize("n,blocks", [(48, 2), (300, 2), (600, 1), (1100, 1), (16, 20)])
def test_bf16_mfma_tower_matches_fp32_oracle(orc, n, blocks):
    """bf16 MFMA path (bf16 weights and inter-layer activations, fp32 accumulate/epilogue) vs the fp32 oracle.
    Stated tolerance: max |dpi| <= 2e-2, max |dv| <= 2e-2; typical error is ~1e-3.
    n selects the 1-, 2- and 3-boards-per-workgroup variants (M = 48 / 96 / 128) and the mixed 3/2 launch."""
    P = pkg()
    base = sample_inputs(64)
    x = np.concatenate([base] * ((n + 63) // 64))[:n].copy()
    flat = T.make_net_flat(blocks, seed=3, perturb_bn=True)
    eng = P.Engine(n, blocks=blocks, sims=1, dtype=P.NET_BF16, node_capacity=64)
    eng.set_weights(flat)
    pi, v = eng.predict(x)
    m = min(n, 64)
    rpi, rv = oracle_forward(orc, flat, blocks, x[:m])
    dpi, dv = np.abs(pi[:m] - rpi).max(), np.abs(v[:m] - rv).max()
    print(f"bf16 n={n} B={blocks}: max|dpi|={dpi:.2e} max|dv|={dv:.2e}")
    assert dpi <= 2e-2 and dv <= 2e-2, (dpi, dv)
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5)
    # identical inputs in different slots / workgroup shapes give identical bits (batch invariance)
    # (n = 600 / 1100 run the mixed launch: 3-board workgroups first, 2-board workgroups after)
    for k in range(64, n - 63, 64):
        assert (pi[:64].view(np.uint32) == pi[k:k + 64].view(np.uint32)).all(), k
        assert (v[:64] == v[k:k + 64]).all(), k
    p1, v1 = eng.predict(x[:3])
    assert (p1.view(np.uint32) == pi[:3].view(np.uint32)).all() and (v1 == v[:3]).all()
    eng.close()


def test_selfplay_bf16_runs_clean():
    """device-resident self-play on the bf16 net: counters consistent, no rule errors, records well-formed"""
    P = pkg()
    G, S = 64, 16
    eng = P.Engine(G, blocks=2, sims=S, dtype=P.NET_BF16, max_game_rounds=34)
    eng.init_random(1)
    eng.selfplay_start(77)
    for _ in range(200):
        eng.selfplay_run(128)
        if eng.counters()["games_finished"] >= G // 2:
            break
    c = eng.counters()
    assert c["errors"] == 0 and c["nodes_dropped"] == 0 and c["games_finished"] >= G // 2
    assert c["simulations"] >= c["decisions"] * S * 0.99
    r = eng.drain()
    assert len(r) == c["samples"] and len(r) > 0
    pi = r[:, 93:265].copy().view(np.float32).reshape(-1, 43)
    z = r[:, 89:93].copy().view(np.float32).reshape(-1)
    assert np.allclose(pi.sum(1), 1.0, atol=1e-4) and set(np.unique(z)) <= {-1.0, 0.0, 1.0}
    assert set(np.unique(r[:, 0])) <= {0, 1}
    eng.close()
