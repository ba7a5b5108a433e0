"""GPU: size-independent properties at BASELINE sizes and edge cases (ragged / empty / maximum inputs) that the
oracle cannot cover at full size in seconds."""
import ctypes as C
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

pytestmark = pytest.mark.gpu
FM = T.data_field_mask()


@pytest.mark.parametrize("G,S,threads,passes", [(256, 100, 1, 450), (256, 100, 2, 300), (512, 100, 2, 200), (2048, 400, 2, 60)])
def test_full_size_selfplay_invariants_and_determinism(orc, G, S, threads, passes):
    """BASELINE configs[1] (G=256, S=100; at THREADS_PER_MCTS 1 and at the bench default 2), the metric's own point
    (G=512, S=100: the bench headline) and configs[2] (G=2048, S=400) shapes, B=20, bf16: two independent engines with the same seeds stay bit-identical (states, RNG streams,
    counters); every exported state satisfies the reference's consistencyCheck identities; evaluation/simulation
    counters balance."""
    P = pkg()
    B = 20
    engs = []
    for _ in range(2):
        e = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=threads)
        e.init_random(20260002)
        e.selfplay_start(20260001)
        e.selfplay_run(passes)
        engs.append(e)
    a, b = engs
    sa, sb = a.get_states(), b.get_states()
    assert (sa == sb).all() and (a.get_rng() == b.get_rng()).all()
    ca, cb = a.counters(), b.counters()
    assert ca == cb
    assert ca["errors"] == 0 and ca["nodes_dropped"] == 0
    # every pass hands one leaf per game and search thread to the net (one per game while a root is being expanded):
    # at T = 1 evaluations consumed = G * (passes - 1) exactly
    assert G * (passes - 1) <= ca["evaluations"] <= G * threads * (passes - 1)
    if threads == 1:
        assert ca["evaluations"] == G * (passes - 1)
    # evaluations = simulations that ended in a leaf + root expansions; terminal simulations need no evaluation
    assert ca["simulations"] >= ca["evaluations"] - ca["decisions"] - G
    assert ca["decisions"] >= G * ((passes - 2) * threads // (S + threads + 1)) - G
    s = T.OrcState()
    back = np.zeros(160, np.uint8)
    for g in range(0, G, 3 if G <= 256 else 17):
        orc.orc_state_unpack(C.byref(s), T.ptr(sa[g]))
        assert orc.orc_consistency_check(C.byref(s)) == 0
        orc.orc_state_pack(C.byref(s), T.ptr(back))
        assert (back == sa[g]).all()
    for e in engs:
        e.close()


def test_import_export_idempotent_and_masks_recomputed():
    g = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    states = g["states"][::11][:512].copy()
    G = len(states)
    eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64)
    eng.set_states(states)
    out1 = eng.get_states()
    assert (out1[:, FM] == states[:, FM]).all()
    # garbage in the derived fields (masks, totalArmy) and in the padding is ignored on import
    dirty = states.copy()
    rng = np.random.default_rng(0)
    for p in range(2):
        b = 48 + 48 * p
        dirty[:, b:b + 40] = rng.integers(0, 256, (G, 40), dtype=np.uint8)
    dirty[:, 42:48] = 0xAA
    dirty[:, 156:160] = 0x55
    eng.set_states(dirty)
    out2 = eng.get_states()
    assert (out2 == out1).all()
    eng.set_states(out2)
    assert (eng.get_states() == out1).all()
    eng.close()


def test_ragged_lockstep_search_finished_games_idle(orc):
    """half of the roots are finished games: they must idle (no evaluation requested, no error) while the others
    complete exactly S simulations each"""
    stub = orc.orc_hash_eval
    stub.argtypes = [C.c_void_p, T.u8p, T.f32p, C.c_void_p]
    g = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    live = g["states"][5::61][:24]
    done = g["finals"][:20]
    states = np.concatenate([live, done])
    G, S = len(states), 40
    eng = pkg().Engine(G, blocks=1, sims=S, dtype=pkg().NET_F32)
    eng.set_states(states)
    eng.set_rng(np.arange(1, G + 1, dtype=np.uint32))
    status = eng.status()
    assert (status[:len(live)] == -1).all() and (status[len(live):] != -1).all()
    eng.mcts_begin()
    pi = np.zeros((G, 43), np.float32); v = np.zeros(G, np.float32); vv = C.c_float(0)
    for _ in range(10000):
        x, need, active = eng.mcts_leaves()
        assert not need[len(live):].any()
        if active == 0:
            break
        for k in np.nonzero(need)[0]:
            stub(None, x[k].ctypes.data_as(T.u8p), pi[k].ctypes.data_as(T.f32p), C.byref(vv)); v[k] = vv.value
        eng.mcts_apply(pi, v)
    n, _, _ = eng.root_stats()
    assert (n[:len(live)].sum(1) == S).all()
    assert (n[len(live):] == 0).all()
    assert (eng.get_states()[:, FM] == states[:, FM]).all()   # a search never mutates the root
    eng.close()


@pytest.mark.parametrize("G", [1, 3])
def test_tiny_engines(G):
    P = pkg()
    eng = P.Engine(G, blocks=1, sims=8, dtype=P.NET_BF16)
    eng.init_random(3)
    eng.new_games(np.arange(1, G + 1, dtype=np.uint32))
    eng.simulate()
    n, _, _ = eng.root_stats()
    assert (n.sum(1) == 8).all()
    eng.selfplay_start(5)
    eng.selfplay_run(200)
    c = eng.counters()
    assert c["errors"] == 0 and c["evaluations"] == G * 199
    eng.close()


def test_node_pool_exhaustion_is_counted_not_fatal():
    """maximum-size stress: a 64-node pool cannot hold a 300-simulation search; insertions are skipped and counted,
    the search still completes S simulations and the game goes on"""
    P = pkg()
    G, S = 32, 300
    eng = P.Engine(G, blocks=1, sims=S, dtype=P.NET_BF16, node_capacity=64)
    eng.init_random(3)
    eng.selfplay_start(9)
    eng.selfplay_run(3 * (S + 40))
    c = eng.counters()
    assert c["nodes_dropped"] > 0 and c["errors"] == 0
    assert c["decisions"] >= G
    eng.close()


def test_seed_streams_are_per_game_and_restart_advances():
    """game g of a slot plays seeds base+g, base+G+g, ... (azr.h); two engines started with bases that differ by G
    see the second engine's first games equal the first engine's second games"""
    P = pkg()
    G = 8
    a = P.Engine(G, blocks=1, sims=2, dtype=P.NET_BF16, max_game_rounds=28)
    a.init_random(1)
    a.selfplay_start(1000)
    b = P.Engine(G, blocks=1, sims=2, dtype=P.NET_BF16, max_game_rounds=28)
    b.init_random(1)
    b.new_games(np.arange(1000, 1000 + G, dtype=np.uint32))
    assert (a.get_states() == b.get_states()).all()
    for _ in range(400):
        a.selfplay_run(50)
        if a.counters()["games_finished"] >= 2 * G:
            break
    assert a.counters()["games_finished"] >= G
    a.close(); b.close()


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_quota_selfplay_is_a_function_of_the_seeds_alone(dtype_name):
    """exactly N games in quota mode (azr_selfplay_start_games: game k plays seed base + k) on engines of 96, 40 and 7 slots: the
    launches differ in every way — leaves per pass from 192 down to 1, so the split-channel tower (<= 128 boards), the one-board
    kernel (129 .. 256) and the emptying tail all run, games finish in another order — but a game is a function of its seed and the
    tower's tiles agree bit for bit, so the multiset of finished records must be the same, byte for byte"""
    P = pkg()
    dt = {"bf16": P.NET_BF16, "f16": P.NET_F16}[dtype_name]
    N, S, blocks = 120, 8, 2
    out = []
    for G in (96, 40, 7):
        eng = P.Engine(G, blocks=blocks, sims=S, dtype=dt, threads=2, max_game_rounds=40)
        eng.init_random(5)
        eng.selfplay_start_games(4242, N)
        recs = []
        for _ in range(4000):
            eng.selfplay_run(64)
            c = eng.counters()
            recs.append(eng.drain())
            if c["games_finished"] >= N:
                break
        c = eng.counters()
        assert c["games_finished"] == N and c["errors"] == 0 and c["records_dropped"] == 0, c
        r = np.concatenate(recs)
        assert len(r) == c["samples"]
        out.append(r[np.lexsort(r.T[::-1])])   # records sorted bytewise
        eng.close()
    assert out[0].shape == out[1].shape == out[2].shape and len(out[0]) > 100
    assert (out[0] == out[1]).all() and (out[0] == out[2]).all()
