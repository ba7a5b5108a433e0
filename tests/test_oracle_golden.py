"""Oracle (oracle/azr_oracle.c) against the committed golden fixtures generated from the real
reference by tests/golden/make_golden.py.  CPU only.  This is what pins the oracle on the GPU box,
where neither /root/reference nor (necessarily) oracle/_ref exist."""
import ctypes as C
import os

import numpy as np

import azr_testlib as T

G = T.GOLDEN
FM = T.data_field_mask()


def load(name):
    return np.load(os.path.join(G, name))


def test_tables(orc):
    g = load("tables.npz")
    buf = (C.c_uint8 * 8)()
    for i in range(42):
        n = orc.orc_neighbours(i, buf)
        assert n == g["neighbour_count"][i]
        assert [buf[j] for j in range(n)] == list(g["neighbours"][i, :n])
        assert orc.orc_neighbour_mask(i) == int(g["neighbour_mask"][i])
    for c in range(7):
        assert orc.orc_continent_mask(c) == int(g["continent_mask"][c])
    for x, y in zip(g["reinf_in"], g["reinf_out"]):
        assert orc.orc_reinforcement_value(int(x)) == int(y)


def test_rng_known_answers(orc):
    g = load("rng_kat.npz")
    r = T.OrcRng()
    for k, s in enumerate(g["seeds"]):
        orc.orc_rng_seed(C.byref(r), int(s))
        assert [orc.orc_rng_dice(C.byref(r)) for _ in range(600)] == list(g["dice"][k])
        orc.orc_rng_seed(C.byref(r), int(s))
        assert [orc.orc_rng_int(C.byref(r)) for _ in range(200)] == list(g["ints"][k])
        orc.orc_rng_seed(C.byref(r), int(s))
        f = np.array([orc.orc_rng_float(C.byref(r)) for _ in range(200)], np.float32)
        assert (f.view(np.uint32) == g["floats"][k].view(np.uint32)).all()
        orc.orc_rng_seed(C.byref(r), int(s))
        for _ in range(10):
            orc.orc_rng_dice(C.byref(r)); orc.orc_rng_int(C.byref(r)); orc.orc_rng_float(C.byref(r))
        assert r.x == int(g["mixed_state"][k])
    orc.orc_rng_seed(C.byref(r), int(g["rm_seed"]))
    out = [orc.orc_random_mask(C.byref(r), int(m)) for m in g["rm_masks"]]
    assert out == [int(x) for x in g["rm_out"]]


def test_rules_games(orc):
    g = load("rules_games.npz")
    for k, seed in enumerate(g["seeds"]):
        lo, hi = g["starts"][k], g["starts"][k + 1]
        h = T.orc_random_game(int(seed))
        assert len(h["moves"]) == hi - lo
        assert (h["moves"] == g["moves"][lo:hi]).all()
        assert (h["masks"] == g["masks"][lo:hi]).all()
        assert (h["states"][:, FM] == g["states"][lo:hi][:, FM]).all()
        assert h["status"] == g["status"][k]
        assert (h["final"][FM] == g["finals"][k][FM]).all()


def test_consistency_invariant_on_golden_states(orc):
    """the reference's consistencyCheck (state.cpp:1181-1429) restated: every derived mask/total of
    every golden state equals a recomputation from landArmy[]"""
    g = load("rules_games.npz")
    s = T.OrcState()
    for st in g["states"][::3]:
        orc.orc_state_unpack(C.byref(s), T.ptr(st))
        assert orc.orc_consistency_check(C.byref(s)) == 0
        back = np.zeros(160, np.uint8)
        orc.orc_state_pack(C.byref(s), T.ptr(back))
        assert (back[FM] == st[FM]).all()


def test_every_move_index_and_error_class(orc):
    g = load("moves_all.npz")
    cfg = T.default_settings()
    s = T.OrcState()
    r = T.OrcRng()
    base = int(g["dice_seed_base"])
    for i, st in enumerate(g["states"]):
        for mv in range(44):
            orc.orc_state_unpack(C.byref(s), T.ptr(st))
            orc.orc_rng_seed(C.byref(r), base + mv)
            rc = orc.orc_make_move(C.byref(s), mv, C.byref(r), C.byref(cfg))
            assert rc == g["rc"][i, mv], (i, mv)
            if rc == 0:
                d = np.zeros(160, np.uint8)
                orc.orc_state_pack(C.byref(s), T.ptr(d))
                assert (d[FM] == g["next"][i, mv][FM]).all(), (i, mv)


def test_encode_and_status(orc):
    g = load("encode.npz")
    cfg = T.default_settings()
    s = T.OrcState()
    out = np.zeros(88, np.uint8)
    for st, ref, gs in zip(g["states"], g["in88"], g["status"]):
        orc.orc_state_unpack(C.byref(s), T.ptr(st))
        orc.orc_encode(C.byref(s), T.ptr(out))
        assert (out == ref).all()
        assert orc.orc_game_status(C.byref(s), C.byref(cfg)) == gs


def test_normalize(orc):
    g = load("normalize.npz")
    for p, vm, ref in zip(g["priors"], g["valid"], g["out"]):
        q = p.copy()
        orc.orc_normalize(T.ptr(q), int(vm))
        assert (q.view(np.uint32) == ref.view(np.uint32)).all()


def test_update_values(orc):
    g = load("update_values.npz")
    pl = g["players"]
    for gs, key in ((0, "z_p0"), (1, "z_p1"), (-2, "z_draw")):
        z = np.zeros(len(pl), np.float32)
        orc.orc_update_values(T.ptr(pl), len(pl), gs, T.ptr(z))
        assert (z == g[key]).all()


def test_planes_follow_plane_indices(orc):
    """setInStateTensor plane order (alphazero_nn_data.h:13-39, V2): parity unpinned (TF unit), checked
    for internal consistency: army planes partition by owner, broadcast planes constant."""
    g = load("encode.npz")
    t = np.zeros((42, 13), np.float32)
    for in88 in g["in88"][::11]:
        orc.orc_planes(T.ptr(in88), T.ptr(t))
        army = (in88[:42] & 63).astype(np.float32) / 32.0
        owner = in88[:42] >> 6
        cur = in88[42]
        assert np.array_equal(t[:, 0], np.where(owner == cur, army, 0))
        assert np.array_equal(t[:, 1], np.where(owner == 1 - cur, army, 0))
        assert np.array_equal(t[:, 2], np.where(owner == 2, army, 0))
        f = in88[48:88].view(np.float32)
        assert (t[:, 3] == f[9]).all() and (t[:, 4] == f[0]).all() and (t[:, 5] == f[1]).all()
        assert (t[:, 6] == f[2]).all()
        assert np.array_equal(t[0, 7:13], f[3:9]) and t[:, 7:13].sum() == 42.0


def test_players_and_game_driver(orc):
    """ScriptPlayer, RandomPlayer and Game's mirrored pairs against the reference's results"""
    g = load("players_games.npz")
    for i in range(len(g["seeds"])):
        k0, k1 = (int(x) for x in g["kinds"][i])
        res, st, rd, fin, rs = T.orc_play_games(k0, k1, 6, int(g["mirror"][i]), int(g["seeds"][i]))
        assert res == tuple(int(x) for x in g["results"][i]), i
        assert (st == g["status"][i]).all() and (rd == g["rounds"][i]).all()
        assert (fin[:, FM] == g["finals"][i][:, FM]).all()
        assert rs == int(g["rng_state"][i])
