"""GPU parity, search rows a15-a24: flattened-array MCTS against the oracle's restatement of
alphazero_mcts.cpp — at one search thread (the reference's only deterministic configuration) and at
THREADS_PER_MCTS = 2 / 4 in the lock-step schedule both sides implement (virtual loss through active_N, duplicate
leaf requests dropped, count = S - S % T).  The net is taken out of the comparison by feeding both sides the
same (pi, v): either a stub evaluated on the host for the device's leaves (azr_mcts_leaves / azr_mcts_apply),
or the device net itself called back from the oracle.  Bit-exact: visit counts N, Q, priors P, policies,
picked moves, next states, RNG streams, (s, pi, z) records.
Parity status: the reference's MCTS units need TensorFlow and cannot be built here, so the oracle side of these
tests is "parity unpinned" (DESIGN.md)."""
import ctypes as C
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

pytestmark = pytest.mark.gpu
FM = T.data_field_mask()


def host_stub_search(eng, orc, stub):
    """azr_mcts_simulate with the NN seam served on the host by an oracle stub net"""
    eng.mcts_begin()
    pi = np.zeros((eng.G * eng.T, 43), np.float32)     # leaf slot = game * T + search thread
    v = np.zeros(eng.G * eng.T, np.float32)
    vv = C.c_float(0)
    for _ in range(100000):
        x, need, active = eng.mcts_leaves()
        if active == 0:
            return
        for g in np.nonzero(need)[0]:
            stub(None, x[g].ctypes.data_as(T.u8p), pi[g].ctypes.data_as(T.f32p), C.byref(vv))
            v[g] = vv.value
        eng.mcts_apply(pi, v)
    raise AssertionError("search did not terminate")


@pytest.mark.parametrize("stub_name,sims,threads", [("orc_hash_eval", 24, 1), ("orc_uniform_eval", 16, 1), ("orc_hash_eval", 100, 1),
                                                    ("orc_hash_eval", 25, 2), ("orc_uniform_eval", 16, 2),
                                                    ("orc_hash_eval", 102, 4), ("orc_uniform_eval", 23, 3)])
def test_search_tree_reuse_and_moves_bit_exact(orc, stub_name, sims, threads):
    """several consecutive decisions per game (tree reuse through trimNodes), positions from all phases"""
    stub = getattr(orc, stub_name)
    stub.argtypes = [C.c_void_p, T.u8p, T.f32p, C.c_void_p]
    gold = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    # starting positions: every 29th golden state (setup, reinforcement, attack, mobilisation, fortify, late game)
    states = gold["states"][::29][:96]
    G = len(states)
    cfg = T.default_settings(mcts_simulations=sims, mcts_threads=threads)
    eng = pkg().Engine(G, blocks=1, sims=sims, dtype=pkg().NET_F32, threads=threads)
    eng.set_states(states)
    seeds = np.arange(500, 500 + G, dtype=np.uint32)
    eng.set_rng(seeds)
    S, R, M = [], [], []
    for g in range(G):
        s, r = T.OrcState(), T.OrcRng()
        orc.orc_state_unpack(C.byref(s), T.ptr(states[g]))
        r.x = int(seeds[g])
        S.append(s); R.append(r); M.append(orc.orc_mcts_create(C.byref(cfg)))
    evalfn = C.cast(stub, C.c_void_p)
    d = np.zeros(160, np.uint8)
    decisions = 6 if sims <= 24 else 3
    for step in range(decisions):
        status = eng.status()
        host_stub_search(eng, orc, stub)
        n_gpu, q_gpu, p_gpu = eng.root_stats()
        pi_gpu = eng.policy()
        sample = step % 2 == 1
        moves = eng.pick(sample=sample)
        moves[status != -1] = 255
        assert (eng.make_moves(moves) == 0).all()
        after, rng_after = eng.get_states(), eng.get_rng()
        n, q, p, pi = (np.zeros(43, np.uint32), np.zeros(43, np.float32), np.zeros(43, np.float32),
                       np.zeros(43, np.float32))
        for g in range(G):
            if status[g] != -1:
                continue
            assert orc.orc_mcts_simulate(M[g], C.byref(S[g]), C.byref(R[g]), evalfn, None) == 0
            orc.orc_mcts_root_stats(M[g], C.byref(S[g]), T.ptr(n), T.ptr(q), T.ptr(p), None)
            assert (n == n_gpu[g]).all(), (step, g, n, n_gpu[g])
            assert (q.view(np.uint32) == q_gpu[g].view(np.uint32)).all(), (step, g)
            assert (p.view(np.uint32) == p_gpu[g].view(np.uint32)).all(), (step, g)
            orc.orc_mcts_policy(M[g], C.byref(S[g]), T.ptr(pi))
            assert (pi.view(np.uint32) == pi_gpu[g].view(np.uint32)).all(), (step, g)
            mv = orc.orc_pick_random(T.ptr(pi), C.byref(R[g])) if sample else orc.orc_pick_highest(T.ptr(pi))
            assert mv == moves[g], (step, g)
            assert orc.orc_make_move(C.byref(S[g]), mv, C.byref(R[g]), C.byref(cfg)) == 0
            orc.orc_state_pack(C.byref(S[g]), T.ptr(d))
            assert (d == after[g]).all(), (step, g)
            assert R[g].x == rng_after[g], (step, g)
    for m in M:
        orc.orc_mcts_destroy(m)
    eng.close()


def test_player_seam_extra_trim_empties_the_tree(orc):
    """AlphaZeroPlayer::takeTurn trims once itself and simulate trims again: after
    azr_mcts_trim the next search starts from an empty table — visit counts equal a fresh search."""
    stub = orc.orc_hash_eval
    stub.argtypes = [C.c_void_p, T.u8p, T.f32p, C.c_void_p]
    G, sims = 16, 16
    eng = pkg().Engine(G, blocks=1, sims=sims, dtype=pkg().NET_F32)
    seeds = np.arange(900, 900 + G, dtype=np.uint32)
    eng.new_games(seeds)
    host_stub_search(eng, orc, stub)
    eng.mcts_trim()              # the player's own trim; the search's trim then drops everything
    rng = eng.get_rng()
    host_stub_search(eng, orc, stub)
    n1, _, _ = eng.root_stats()
    eng.mcts_clear()
    eng.set_rng(rng)
    host_stub_search(eng, orc, stub)
    n2, _, _ = eng.root_stats()
    assert (n1 == n2).all() and (n1.sum(1) == sims).all()
    eng.close()


@pytest.mark.parametrize("threads", [1, 2])
def test_full_selfplay_games_device_resident_vs_oracle(orc, threads):
    """the device-resident trainer loop (azr_selfplay_run: search, temperature pick, record, move, z back-fill,
    restart) against orc_selfplay_game with the DEVICE net called back for every evaluation: identical record
    streams (265-byte layout), move for move."""
    G, sims, B = 6, 6, 1
    P = pkg()
    eng = P.Engine(G, blocks=B, sims=sims, dtype=P.NET_F32, max_game_rounds=36, threads=threads)
    flat = T.make_net_flat(B, seed=11, perturb_bn=True)
    eng.set_weights(flat)
    base = 4242
    eng.selfplay_start(base)
    recs = []
    for _ in range(400):
        eng.selfplay_run(64)
        c = eng.counters()
        if c["games_finished"] >= G:
            break
    c = eng.counters()
    assert c["games_finished"] >= G and c["errors"] == 0 and c["nodes_dropped"] == 0
    recs = eng.drain()
    assert len(recs) == c["samples"]

    @T.EVAL_FN
    def hip_eval(ctx, in88, pi, v):
        x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
        p, vv = eng.predict(x)
        C.memmove(pi, p.ctypes.data, 43 * 4)
        v[0] = float(vv[0])

    cfg = T.default_settings(mcts_simulations=sims, max_game_rounds=36, mcts_threads=threads)
    # records of one game are contiguous in the ring; games finish in any order: match by content
    want = {}
    for g in range(G):
        buf = np.zeros((4096, 265), np.uint8)
        st, rounds = C.c_int(0), C.c_int(0)
        n = orc.orc_selfplay_game(C.byref(cfg), base + g, hip_eval, None, T.ptr(buf), 4096, C.byref(st),
                                  C.byref(rounds), None, 0, None, None)
        assert n > 0
        want[g] = buf[:n].copy()
    blob = recs.tobytes()
    for g in range(G):
        assert want[g].tobytes() in blob, f"game {g} record stream not found in the device's output"
    eng.close()


def test_selfplay_from_midgame_states_vs_oracle(orc):
    """azr_selfplay_start_from_states: the trainer's move loop entered in the MIDDLE of games (bench.py's second timed leg).
    12 golden positions of all phases, their own RNG streams; the device plays every game to its end and must produce the
    record stream of the oracle's loop (search, temperature pick, record, move — orc_selfplay_game's body) started from the
    same position with the same stream, the DEVICE net called back for every evaluation."""
    P = pkg()
    sims, B = 6, 1
    gold = np.load(os.path.join(T.GOLDEN, "rules_games.npz"))
    states = gold["states"][::471][:12].copy()
    G = len(states)
    assert len(set(states[:, 149])) >= 4   # several phases
    seeds = np.arange(7000, 7000 + G, dtype=np.uint32)
    eng = P.Engine(G, blocks=B, sims=sims, dtype=P.NET_F32, threads=1)
    flat = T.make_net_flat(B, seed=11, perturb_bn=True)
    eng.set_weights(flat)
    eng.set_states(states)
    eng.set_rng(seeds)
    eng.selfplay_start_from_states(31337)
    for _ in range(400):
        eng.selfplay_run(64)
        if eng.counters()["games_finished"] >= 3 * G:   # every slot's FIRST game (the continued one) is over long before
            break
    c = eng.counters()
    assert c["errors"] == 0 and c["nodes_dropped"] == 0 and c["records_dropped"] == 0
    blob = eng.drain().tobytes()

    @T.EVAL_FN
    def hip_eval(ctx, in88, pi, v):
        x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
        p, vv = eng.predict(x)
        C.memmove(pi, p.ctypes.data, 43 * 4)
        v[0] = float(vv[0])

    cfg = T.default_settings(mcts_simulations=sims, mcts_threads=1)
    for g in range(G):
        s, r = T.OrcState(), T.OrcRng()
        orc.orc_state_unpack(C.byref(s), T.ptr(states[g]))
        r.x = int(seeds[g])
        m = orc.orc_mcts_create(C.byref(cfg))
        recs, players = [], []
        gs = orc.orc_game_status(C.byref(s), C.byref(cfg))
        assert gs == -1
        while gs == -1:
            assert orc.orc_mcts_simulate(m, C.byref(s), C.byref(r), hip_eval, None) == 0
            pi = np.zeros(43, np.float32)
            assert orc.orc_mcts_policy(m, C.byref(s), T.ptr(pi)) == 0
            mv = orc.orc_pick_highest(T.ptr(pi)) if s.round > cfg.temperature_threshold else orc.orc_pick_random(T.ptr(pi), C.byref(r))
            rec = np.zeros(265, np.uint8)
            rec[0] = np.uint8(s.cur)
            orc.orc_encode(C.byref(s), T.ptr(rec[1:89]))
            rec[93:265] = pi.view(np.uint8)
            recs.append(rec)
            players.append(s.cur)
            assert orc.orc_make_move(C.byref(s), mv, C.byref(r), C.byref(cfg)) == 0
            gs = orc.orc_game_status(C.byref(s), C.byref(cfg))
        orc.orc_mcts_destroy(m)
        z = np.zeros(len(recs), np.float32)
        orc.orc_update_values(T.ptr(np.array(players, np.int8)), len(recs), gs, T.ptr(z))
        for i, rec in enumerate(recs):
            rec[89:93] = z[i:i + 1].view(np.uint8)
        want = np.stack(recs).tobytes()
        assert want in blob, f"slot {g}: the continued game's record stream was not found in the device's output"
    eng.close()
