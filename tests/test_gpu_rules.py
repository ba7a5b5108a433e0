"""GPU parity, rules rows a1-a12: the wave-resident state-step kernels, called through the C-ABI,
against the oracle and the committed golden fixtures.  Bit-exact: states (160-B images incl. the derived
masks), legal masks, dice/RNG streams, outcomes, error classes, feature structs."""
import ctypes as C
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

pytestmark = pytest.mark.gpu
FM = T.data_field_mask()


def load(name):
    return np.load(os.path.join(T.GOLDEN, name))


def test_new_games_bit_exact(orc):
    G = 512
    eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64)
    seeds = np.concatenate([np.arange(1, G - 2, dtype=np.uint32), np.array([0, 2147483647, 4294967295], np.uint32)])
    eng.new_games(seeds)
    got, rng = eng.get_states(), eng.get_rng()
    s, r, d = T.OrcState(), T.OrcRng(), np.zeros(160, np.uint8)
    for g in range(G):
        orc.orc_rng_seed(C.byref(r), int(seeds[g]))
        orc.orc_new_game(C.byref(s), C.byref(r))
        orc.orc_state_pack(C.byref(s), T.ptr(d))
        assert (d == got[g]).all(), g
        assert r.x == rng[g]
    eng.close()


def test_lockstep_random_games_vs_oracle_and_golden(orc):
    """G games stepped to the end in lock-step; move choice = the oracle's randomMask on its own stream (exactly the
    golden generator's policy), dice = the device's per-game minstd_rand0 stream kept aligned with the oracle's."""
    gold = load("rules_games.npz")
    cfg = T.default_settings()
    seeds = np.concatenate([gold["seeds"], np.arange(3000, 3000 + 236, dtype=np.uint32)])
    G = len(seeds)
    eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64)
    eng.new_games(seeds)
    S = [T.OrcState() for _ in range(G)]
    R = [T.OrcRng() for _ in range(G)]
    for g in range(G):
        orc.orc_rng_seed(C.byref(R[g]), int(seeds[g]))
        orc.orc_new_game(C.byref(S[g]), C.byref(R[g]))
    traj = [[] for _ in range(G)]
    d = np.zeros(160, np.uint8)
    steps = 0
    while True:
        status = eng.status()
        vm = eng.valid_moves()
        states = eng.get_states()
        moves = np.full(G, 255, np.uint8)
        rng = np.zeros(G, np.uint32)
        live = 0
        for g in range(G):
            st = orc.orc_game_status(C.byref(S[g]), C.byref(cfg))
            assert st == status[g], (g, steps)
            orc.orc_state_pack(C.byref(S[g]), T.ptr(d))
            assert (d == states[g]).all(), (g, steps)
            if st != -1:
                rng[g] = R[g].x
                continue
            live += 1
            m = orc.orc_valid_moves(C.byref(S[g]), C.byref(cfg))
            assert m == int(vm[g]), (g, steps, hex(m), hex(int(vm[g])))
            mv = int(orc.orc_random_mask(C.byref(R[g]), m)).bit_length() - 1
            moves[g] = mv
            traj[g].append(mv)
            rng[g] = R[g].x
        if live == 0:
            break
        eng.set_rng(rng)
        rc = eng.make_moves(moves)
        assert (rc == 0).all()
        for g in range(G):
            if moves[g] != 255:
                assert orc.orc_make_move(C.byref(S[g]), int(moves[g]), C.byref(R[g]), C.byref(cfg)) == 0
        assert (eng.get_rng()[moves != 255] == np.array([R[g].x for g in range(G)], np.uint32)[moves != 255]).all(), steps
        steps += 1
        assert steps < 5000
    # the first len(gold seeds) games are the golden ones: same move lists and outcomes as the REFERENCE produced
    final = eng.get_states()
    for k in range(len(gold["seeds"])):
        lo, hi = gold["starts"][k], gold["starts"][k + 1]
        assert traj[k] == list(gold["moves"][lo:hi])
        assert eng.status()[k] == gold["status"][k]
        assert (final[k][FM] == gold["finals"][k][FM]).all()
    eng.close()


def test_every_move_index_error_class_and_next_state():
    g = load("moves_all.npz")
    states = np.repeat(g["states"], 44, axis=0)
    moves = np.tile(np.arange(44, dtype=np.uint8), len(g["states"]))
    G = len(states)
    eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64)
    eng.set_states(states)
    # dice seed = base + move, as the generator seeded the reference's engine before each call
    seeds = (int(g["dice_seed_base"]) + moves.astype(np.uint32)) % 2147483647
    eng.set_rng(seeds.astype(np.uint32))
    rc = eng.make_moves(moves)
    after = eng.get_states()
    assert (rc.reshape(-1, 44) == g["rc"]).all()
    ok = (g["rc"] == 0).reshape(-1)
    assert (after[ok][:, FM] == g["next"].reshape(-1, 160)[ok][:, FM]).all()
    # a throwing move leaves the stored game untouched
    assert (after[~ok][:, FM] == states[~ok][:, FM]).all()
    eng.close()


def test_encode_status_and_roundtrip():
    g = load("encode.npz")
    G = len(g["states"])
    eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64)
    eng.set_states(g["states"])
    assert (eng.encode() == g["in88"]).all()
    assert (eng.status() == g["status"]).all()
    # import ignores the image's derived masks and export recomputes them: must reproduce the reference's
    assert (eng.get_states()[:, FM] == g["states"][:, FM]).all()
    eng.close()


def test_rule_switches_masks(orc):
    """legal masks under non-default LIMIT_* settings against the oracle (pinned to the reference in
    tests/test_oracle_vs_ref.py::test_rule_switches)"""
    g = load("rules_games.npz")
    states = g["states"][::7]
    G = len(states)
    s = T.OrcState()
    for kw in (dict(limit_reinforcement=0), dict(limit_attack=1), dict(allow_yield=0, max_game_rounds=40)):
        cfg = T.default_settings(**kw)
        eng = pkg().Engine(G, blocks=1, sims=1, dtype=pkg().NET_F32, node_capacity=64, **kw)
        eng.set_states(states)
        vm, st = eng.valid_moves(), eng.status()
        for i in range(G):
            orc.orc_state_unpack(C.byref(s), T.ptr(states[i]))
            assert orc.orc_valid_moves(C.byref(s), C.byref(cfg)) == int(vm[i])
            assert orc.orc_game_status(C.byref(s), C.byref(cfg)) == st[i]
        eng.close()
