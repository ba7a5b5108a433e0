"""GPU parity, SURVEY §8 f-1 / f-3 and BASELINE configs[0]: the device arena (GameGroup::playGames with
AlphaZeroPlayer / ScriptPlayer / RandomPlayer, mirrored pairs) against the oracle, which is pinned bit-exactly to the
real reference for ScriptPlayer, RandomPlayer and the Game driver (tests/test_oracle_vs_ref.py)."""
import ctypes as C

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

pytestmark = pytest.mark.gpu
FM = T.data_field_mask()


def run_arena(eng, k0, k1, total, cap, mirror, base):
    eng.arena_start(k0, k1, total, per_slot_cap=cap, mirror=mirror, base_seed=base)
    for _ in range(2000):
        if eng.arena_run(64):
            break
    else:
        raise AssertionError("arena did not finish")
    return eng.arena_results(), eng.arena_log()


@pytest.mark.parametrize("kinds", [(1, 2), (2, 1), (1, 1), (2, 2)])
@pytest.mark.parametrize("mirror", [True, False])
def test_script_and_random_players_arena_bit_exact(orc, kinds, mirror):
    P = pkg()
    G, per_slot, base = 96, 6, 555
    eng = P.Engine(G, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    res, (n, st, rd, fin) = run_arena(eng, kinds[0], kinds[1], 10 ** 6, per_slot, mirror, base)
    assert (n == per_slot).all() and eng.counters()["errors"] == 0
    tot = np.zeros(6, np.int64)
    for g in range(G):
        r6, ost, ord_, ofin, _ = T.orc_play_games(kinds[0], kinds[1], per_slot, mirror, base + g)
        assert (st[g, :per_slot] == ost).all(), g
        assert (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
    assert [res["count"], res["draw"], res["win"][0], res["win_and_started"][0], res["win"][1],
            res["win_and_started"][1]] == list(tot)
    eng.close()


def test_ten_thousand_games_through_the_device_rules_bit_exact(orc):
    """volume check of the device rules engine (SURVEY §7 gate 4 at the size of gate 2): 1024 slots x 10 games of
    RandomPlayer vs ScriptPlayer, unmirrored (every game a fresh deal): statuses, round counts, final states per game"""
    P = pkg()
    G, per_slot, base = 1024, 10, 900000
    eng = P.Engine(G, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    res, (n, st, rd, fin) = run_arena(eng, P.PLAYER_RANDOM, P.PLAYER_SCRIPT, 10 ** 7, per_slot, False, base)
    assert (n == per_slot).all() and eng.counters()["errors"] == 0 and res["count"] == G * per_slot
    tot = np.zeros(6, np.int64)
    for g in range(G):
        r6, ost, ord_, ofin, _ = T.orc_play_games(P.PLAYER_RANDOM, P.PLAYER_SCRIPT, per_slot, False, base + g)
        assert (st[g, :per_slot] == ost).all() and (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
    assert [res["count"], res["draw"], res["win"][0], res["win_and_started"][0], res["win"][1],
            res["win_and_started"][1]] == list(tot)
    eng.close()


def test_counter_semantics_pairs_and_quota():
    """Counter::hasNext(2) (game.cpp:14-26): games are taken in pairs from a shared counter; an odd quota leaves the
    last game unplayed"""
    P = pkg()
    eng = P.Engine(8, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    res, (n, _, _, _) = run_arena(eng, P.PLAYER_SCRIPT, P.PLAYER_RANDOM, 21, 0, True, 1)
    assert res["count"] == 20 and n.sum() == 20 and (n % 2 == 0).all()
    assert res["draw"] + res["win"][0] + res["win"][1] == 20
    eng.close()


@pytest.mark.parametrize("az_first", [True, False])
def test_alphazero_vs_script_config0_bit_exact(orc, az_first):
    """BASELINE configs[0] shape: `-m play --mcts=16` AlphaZero vs ScriptPlayer.  Oracle side: AlphaZeroPlayer::takeTurn
    (extra trim per turn, argmax) with the DEVICE net called back for every evaluation."""
    P = pkg()
    G, per_slot, S, B, base = 6, 2, 16, 1, 4100
    eng = P.Engine(G, blocks=B, sims=S, dtype=P.NET_F32)
    eng.set_weights(T.make_net_flat(B, seed=21, perturb_bn=True))
    k = (P.PLAYER_ALPHAZERO, P.PLAYER_SCRIPT) if az_first else (P.PLAYER_SCRIPT, P.PLAYER_ALPHAZERO)
    res, (n, st, rd, fin) = run_arena(eng, k[0], k[1], 10 ** 6, per_slot, True, base)
    assert (n == per_slot).all() and eng.counters()["errors"] == 0

    @T.EVAL_FN
    def hip_eval(ctx, in88, pi, v):
        x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
        p, vv = eng.predict(x)
        C.memmove(pi, p.ctypes.data, 43 * 4)
        v[0] = float(vv[0])

    cfg = T.default_settings(mcts_simulations=S)
    tot = np.zeros(6, np.int64)
    for g in range(G):
        r6, ost, ord_, ofin, _ = T.orc_play_games(k[0], k[1], per_slot, True, base + g, cfg=cfg, eval_fn=hip_eval)
        assert (st[g, :per_slot] == ost).all(), (g, st[g], ost)
        assert (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
    assert res["count"] == tot[0] and res["win"][0] == tot[2] and res["win"][1] == tot[4]
    eng.close()


@pytest.mark.parametrize("threads,b_first,sb", [(1, False, None), (2, False, None), (2, True, None), (2, False, "2"), (2, False, "3"), (2, True, "4")])
def test_two_net_arena_bit_exact_with_samples(orc, monkeypatch, threads, b_first, sb):
    """New-vs-old arena of the learn loop (GameGroup::playGames(trainAZPG, generateAZPG, ..., trainStorage),
    alphazero_trainer.cpp:143-152): two AlphaZero players with their own trees and DIFFERENT networks in every slot, each
    network evaluated on its own player's leaves only, (s, pi, z) records collected from both players.  Oracle side: the
    same game driver with the two DEVICE nets called back per evaluation."""
    P = pkg()
    G, per_slot, S, B, base = 6, 2, 12, 1, 5200
    if sb:   # every net launch on the single-image tiles ("2": 4 boards per workgroup, "3": 2, "4": 3), through the leaf-slot map
        monkeypatch.setenv("AZR_TOWER_SB", sb)
    a = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=threads, max_game_rounds=40, test_hooks=bool(sb))
    b = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=threads, max_game_rounds=40, test_hooks=bool(sb))
    a.set_weights(T.make_net_flat(B, seed=31, perturb_bn=True))
    b.set_weights(T.make_net_flat(B, seed=32, perturb_bn=True))
    a.arena_set_opponent(b)
    a.arena_collect_samples(True)
    k = (P.PLAYER_ALPHAZERO_B, P.PLAYER_ALPHAZERO) if b_first else (P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B)
    res, (n, st, rd, fin) = run_arena(a, k[0], k[1], 10 ** 6, per_slot, True, base)
    assert (n == per_slot).all() and a.counters()["errors"] == 0 and a.counters()["nodes_dropped"] == 0
    recs = a.drain()

    def make_eval(eng):
        @T.EVAL_FN
        def f(ctx, in88, pi, v):
            x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
            p, vv = eng.predict(x)
            C.memmove(pi, p.ctypes.data, 43 * 4)
            v[0] = float(vv[0])
        return f

    ea, eb = make_eval(a), make_eval(b)
    cfg = T.default_settings(mcts_simulations=S, mcts_threads=threads, max_game_rounds=40)
    tot = np.zeros(6, np.int64)
    blob = recs.tobytes()
    nrec = 0
    for g in range(G):
        r6, ost, ord_, ofin, orec = T.orc_play_games2(k[0], k[1], per_slot, True, base + g, cfg, ea, eb)
        assert (st[g, :per_slot] == ost).all(), (g, st[g], ost)
        assert (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
        for gi, game in enumerate(orec):   # a finished game's records are flushed contiguously, z filled in
            assert len(game) > 0 and set(np.unique(game[:, 0])) == {0, 1}, (g, gi)   # both players contributed
            z = game[:, 89:93].copy().view(np.float32).reshape(-1)
            want_z = np.where(ost[gi] == -2, 0.0, np.where(game[:, 0] == ost[gi], 1.0, -1.0))
            assert (z == want_z).all(), (g, gi)
            assert game.tobytes() in blob, (g, gi)
            nrec += len(game)
    assert nrec == len(recs)
    assert [res["count"], res["draw"], res["win"][0], res["win_and_started"][0], res["win"][1], res["win_and_started"][1]] == list(tot)
    a.arena_set_opponent(None)
    a.close(); b.close()


# ---- AZR_MIRROR_CONCURRENT: the two games of a mirrored pair at the same time on slots 2j and 2j + 1 (include/azr.h) -------------
def _half_slot(g, G, base):
    """slot g of a G-slot concurrent arena = (half, first pair seed, pair stride) of tests/azr_testlib.orc_play_half_games"""
    return g & 1, base + (g >> 1), G // 2


@pytest.mark.parametrize("kinds", [(1, 2), (2, 1), (1, 1)])
def test_concurrent_pair_halves_script_and_random_bit_exact(orc, kinds):
    """every slot against the oracle's slot of the same (half, pair seeds) — whose half 0 is anchored on the real reference
    (tests/test_oracle_vs_ref.py::test_concurrent_halves_anchor_on_the_reference) — and the quota form: pairs assigned statically"""
    P = pkg()
    G, per_slot, base = 64, 4, 777
    eng = P.Engine(G, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    res, (n, st, rd, fin) = run_arena(eng, kinds[0], kinds[1], 10 ** 6, per_slot, P.MIRROR_CONCURRENT, base)
    assert (n == per_slot).all() and eng.counters()["errors"] == 0
    tot = np.zeros(6, np.int64)
    for g in range(G):
        half, q0, stride = _half_slot(g, G, base)
        r6, ost, ord_, ofin, _ = T.orc_play_half_games(kinds[0], kinds[1], per_slot, half, q0, stride)
        assert (st[g, :per_slot] == ost).all() and (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
    assert [res["count"], res["draw"], res["win"][0], res["win_and_started"][0], res["win"][1], res["win_and_started"][1]] == list(tot)
    # quota: 37 games = 18 pairs over 32 slot pairs -> slot pairs 0..17 play one pair each, the others nothing; G odd leaves the last slot idle
    res, (n, _, _, _) = run_arena(eng, kinds[0], kinds[1], 37, 0, P.MIRROR_CONCURRENT, base)
    assert res["count"] == 36 and (n[:36] == 1).all() and (n[36:] == 0).all()
    eng.close()
    odd = P.Engine(5, blocks=1, sims=1, dtype=P.NET_F32, node_capacity=64)
    res, (n, _, _, _) = run_arena(odd, kinds[0], kinds[1], 12, 0, P.MIRROR_CONCURRENT, base)
    assert res["count"] == 12 and list(n) == [3, 3, 3, 3, 0]
    odd.close()


@pytest.mark.parametrize("threads,b_first", [(2, False), (1, True)])
def test_concurrent_pair_halves_two_net_arena_bit_exact_with_samples(orc, threads, b_first):
    """the learn loop's new-vs-old comparison in the concurrent form: results, final states, rounds and every (s, pi, z) record of
    both AlphaZero players against the oracle's slots, the two DEVICE nets called back per evaluation"""
    P = pkg()
    G, per_slot, S, B, base = 6, 2, 12, 1, 6100
    a = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=threads, max_game_rounds=40)
    b = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=threads, max_game_rounds=40)
    a.set_weights(T.make_net_flat(B, seed=31, perturb_bn=True))
    b.set_weights(T.make_net_flat(B, seed=32, perturb_bn=True))
    a.arena_set_opponent(b)
    a.arena_collect_samples(True)
    k = (P.PLAYER_ALPHAZERO_B, P.PLAYER_ALPHAZERO) if b_first else (P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B)
    res, (n, st, rd, fin) = run_arena(a, k[0], k[1], 10 ** 6, per_slot, P.MIRROR_CONCURRENT, base)
    assert (n == per_slot).all() and a.counters()["errors"] == 0 and a.counters()["nodes_dropped"] == 0
    recs = a.drain()

    def make_eval(eng):
        @T.EVAL_FN
        def f(ctx, in88, pi, v):
            x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
            p, vv = eng.predict(x)
            C.memmove(pi, p.ctypes.data, 43 * 4)
            v[0] = float(vv[0])
        return f

    ea, eb = make_eval(a), make_eval(b)
    cfg = T.default_settings(mcts_simulations=S, mcts_threads=threads, max_game_rounds=40)
    tot = np.zeros(6, np.int64)
    blob = recs.tobytes()
    nrec = 0
    for g in range(G):
        half, q0, stride = _half_slot(g, G, base)
        r6, ost, ord_, ofin, orec = T.orc_play_half_games(k[0], k[1], per_slot, half, q0, stride, cfg, ea, eb)
        assert (st[g, :per_slot] == ost).all(), (g, st[g], ost)
        assert (rd[g, :per_slot] == ord_).all(), g
        assert (fin[g, :per_slot][:, FM] == ofin[:, FM]).all(), g
        tot += np.array(r6)
        for gi, game in enumerate(orec):
            assert len(game) > 0 and game.tobytes() in blob, (g, gi)
            nrec += len(game)
    assert nrec == len(recs)
    assert [res["count"], res["draw"], res["win"][0], res["win_and_started"][0], res["win"][1], res["win_and_started"][1]] == list(tot)
    a.arena_set_opponent(None)
    a.close(); b.close()


# ---- passes queued without a read-back (azr_arena_run, up to 256 leaf slots on the 16-bit towers) ---------------------------------
@pytest.mark.parametrize("two_nets", [False, True])
def test_passes_without_a_read_back_equal_the_read_back_form(monkeypatch, two_nets):
    """The net launches of an arena of <= 256 leaf slots read the tree step's leaf count from device memory (net_forward_counted): the
    same arena through the read-back form (AZR_ARENA_COUNTED=0, test build) must give the same games and records.  128 slots x T = 2
    with the AlphaZero player first: the opening passes carry 256 leaves for one net — more than the split-channel tower takes; the
    launch says so in its give-up word (value 2) and the one-board-per-workgroup launch behind it computes them, not counted as a
    hand-off that gave up."""
    P = pkg()
    G, S, B, base = 128, 6, 1, 9100
    out = []
    for counted in (True, False):
        if not counted:
            monkeypatch.setenv("AZR_ARENA_COUNTED", "0")
        a = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=2, max_game_rounds=30, test_hooks=not counted)
        a.set_weights(T.make_net_flat(B, seed=41, perturb_bn=True))
        b = None
        if two_nets:
            b = P.Engine(G, blocks=B, sims=S, dtype=P.NET_BF16, threads=2, max_game_rounds=30, test_hooks=not counted)
            b.set_weights(T.make_net_flat(B, seed=42, perturb_bn=True))
            a.arena_set_opponent(b)
        a.arena_collect_samples(True)
        k = (P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B) if two_nets else (P.PLAYER_ALPHAZERO, P.PLAYER_SCRIPT)
        res, (n, st, rd, fin) = run_arena(a, k[0], k[1], 256, 0, P.MIRROR_SEQUENTIAL, base)
        c = a.counters()
        assert c["errors"] == 0 and c["nodes_dropped"] == 0 and c["tower_fallbacks"] == 0 and c["records_dropped"] == 0
        recs = a.drain()
        out.append((res, n.copy(), st.copy(), rd.copy(), fin.copy(), b"".join(sorted(r.tobytes() for r in recs))))
        if two_nets:
            a.arena_set_opponent(None)
            b.close()
        a.close()
    assert out[0][0] == out[1][0] and out[0][0]["count"] == 256
    for x, y in zip(out[0][1:5], out[1][1:5]):
        assert (x == y).all()
    assert len(out[0][5]) > 0 and out[0][5] == out[1][5]
