"""Live pinning of the oracle against the REAL reference (oracle/_ref/libazr_ref.so, compiled from
/root/reference by oracle/Makefile).  Skipped where oracle/_ref is absent."""
import ctypes as C

import numpy as np
import pytest

import azr_testlib as T

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref not built")]
FM = T.data_field_mask()


def test_bulk_random_games_bit_exact(orc):
    steps = 0
    for seed in range(1000, 11000):   # SURVEY §7 gate 2: >= 10 k seeded games, every state / mask / move / outcome
        g = T.ref_random_game(seed)
        h = T.orc_random_game(seed)
        assert len(g["moves"]) == len(h["moves"]), seed
        assert (g["moves"] == h["moves"]).all(), seed
        assert (g["masks"] == h["masks"]).all(), seed
        assert (g["states"][:, FM] == h["states"][:, FM]).all(), seed
        assert g["status"] == h["status"] and (g["final"][FM] == h["final"][FM]).all()
        steps += len(g["moves"])
    assert steps > 3000000


@pytest.mark.parametrize("rules", [dict(allow_yield=0), dict(limit_reinforcement=0), dict(limit_attack=1),
                                   dict(max_game_rounds=40), dict(min_unit_move=1)])
def test_rule_switches(orc, rules):
    base = dict(allow_yield=1, limit_reinforcement=1, limit_attack=0, max_game_rounds=58, min_unit_move=3)
    base.update(rules)
    R = T.ref()
    R.ref_set_rules(base["allow_yield"], base["limit_reinforcement"], base["limit_attack"], base["max_game_rounds"],
                    base["min_unit_move"])
    try:
        cfg = T.default_settings(**base)
        for seed in range(50, 90):
            g = T.ref_random_game(seed)
            h = T.orc_random_game(seed, cfg=cfg)
            assert len(g["moves"]) == len(h["moves"]) and (g["moves"] == h["moves"]).all(), (rules, seed)
            assert (g["masks"] == h["masks"]).all() and g["status"] == h["status"]
            assert (g["final"][FM] == h["final"][FM]).all()
    finally:
        R.ref_set_rules(1, 1, 0, 58, 3)


def test_encode_bit_exact(orc):
    R = T.ref()
    s = T.OrcState()
    for seed in (5, 6, 7):
        g = T.ref_random_game(seed)
        for st in g["states"][::3]:
            a = np.zeros(88, np.uint8); b = np.zeros(88, np.uint8)
            R.ref_encode(T.ptr(st), T.ptr(a))
            orc.orc_state_unpack(C.byref(s), T.ptr(st))
            orc.orc_encode(C.byref(s), T.ptr(b))
            assert (a == b).all()


def test_invert_players(orc):
    R = T.ref()
    s = T.OrcState()
    g = T.ref_random_game(11)
    for st in g["states"][::9]:
        a = st.copy()
        R.ref_invert_players(T.ptr(a))
        orc.orc_state_unpack(C.byref(s), T.ptr(st))
        orc.orc_invert_players(C.byref(s))
        b = np.zeros(160, np.uint8)
        orc.orc_state_pack(C.byref(s), T.ptr(b))
        assert (a[FM] == b[FM]).all()


@pytest.mark.parametrize("kinds", [(1, 2), (2, 1), (1, 1), (2, 2)])
def test_players_and_game_driver_bit_exact(orc, kinds):
    """ScriptPlayer / RandomPlayer / Game::playGames (mirrored pairs, alternating starts) against the real reference:
    results, every final state, round counts and the RNG stream position"""
    for mirror in (1, 0):
        for seed in range(100, 130):
            a = T.ref_play_games(kinds[0], kinds[1], 8, mirror, seed)
            b = T.orc_play_games(kinds[0], kinds[1], 8, mirror, seed)
            assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()
            assert (a[3][:, FM] == b[3][:, FM]).all() and a[4] == b[4]


@pytest.mark.parametrize("kinds", [(1, 2), (2, 1), (1, 1)])
def test_concurrent_halves_anchor_on_the_reference(orc, kinds):
    """AZR_MIRROR_CONCURRENT (include/azr.h): half 0 of the pair with seed q IS the first game a reference thread plays on a
    global engine seeded q (deal and dice from one stream) — results, final state, rounds, RNG position; half 1 is a game of its own
    (mirrored deal, player 1 starts, own dice stream) and a function of (q, half) alone."""
    for q in range(300, 320):
        a = T.ref_play_games(kinds[0], kinds[1], 1, 1, q)
        b = T.orc_play_half_games(kinds[0], kinds[1], 1, 0, q, 1)
        assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all() and (a[3][:, FM] == b[3][:, FM]).all()
        h1 = T.orc_play_half_games(kinds[0], kinds[1], 1, 1, q, 1)
        h1b = T.orc_play_half_games(kinds[0], kinds[1], 3, 1, q - 14, 7)   # the same pair as the third game of another slot
        assert h1[0][0] == 1 and h1[1][0] == h1b[1][2] and (h1[3][0][FM] == h1b[3][2][FM]).all()
        assert h1[0][3] == 0 and h1[0][5] == h1[0][4]   # player 1 started: only its wins count as "won and started"
