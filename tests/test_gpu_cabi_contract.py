"""GPU: error contract and lifetime behaviour of the C-ABI (include/azr.h) — misuse returns AZR_E_* codes with a message,
never crashes; handles can be created and destroyed repeatedly without leaking device memory."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_common import pkg

pytestmark = pytest.mark.gpu


def test_misuse_returns_error_codes(tmp_path):
    P = pkg()
    L = P.load_library()
    eng = P.Engine(4, blocks=1, sims=4, dtype=P.NET_BF16, node_capacity=64)
    # no weights yet: every path that needs the net refuses (AZR_E_STATE = 7)
    for call in (eng.simulate, lambda: eng.predict(np.zeros((1, 88), np.uint8)), lambda: eng.selfplay_run(1)):
        with pytest.raises(P.AzrError) as e:
            call()
        assert e.value.code == 7
    with pytest.raises(P.AzrError) as e:
        eng.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_SCRIPT, 4)
    assert e.value.code == 7
    # wrong parameter count / missing or foreign checkpoint files
    with pytest.raises(P.AzrError) as e:
        eng.set_weights(np.zeros(10, np.float32))
    assert e.value.code == 1
    with pytest.raises(P.AzrError) as e:
        eng.load(str(tmp_path / "nope.bin"))
    assert e.value.code == 6
    (tmp_path / "junk.bin").write_bytes(b"not a checkpoint" * 10)
    with pytest.raises(P.AzrError) as e:
        eng.load(str(tmp_path / "junk.bin"))
    assert e.value.code == 6
    other = P.Engine(2, blocks=2, sims=1, dtype=P.NET_F32, node_capacity=64)
    other.init_random(1)
    other.save(str(tmp_path / "b2.bin"))
    with pytest.raises(P.AzrError):      # block count mismatch
        eng.load(str(tmp_path / "b2.bin"))
    other.close()
    # null / bad arguments through the raw ABI
    assert L.azr_engine_new_games(eng.h, None) == 1
    assert L.azr_engine_create(None, None) == 1
    assert L.azr_engine_games(None) == 0
    assert L.azr_mcts_simulate(None) == 3          # AZR_E_BAD_HANDLE
    # the engine is still usable afterwards
    eng.init_random(3)
    eng.new_games(np.arange(1, 5, dtype=np.uint32))
    eng.simulate()
    assert (eng.root_stats()[0].sum(1) == 4).all()
    # arena kinds: AlphaZero vs AlphaZero in one engine is refused with a clear message
    with pytest.raises(P.AzrError) as e:
        eng.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO, 4)
    assert "one tree each" in str(e.value)
    # ... the opponent player without an opponent handle too
    with pytest.raises(P.AzrError) as e:
        eng.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B, 4)
    assert e.value.code == 7 and "azr_arena_set_opponent_net" in str(e.value)
    eng.close()


def test_thread_count_is_validated():
    P = pkg()
    for kw in (dict(threads=0), dict(threads=9), dict(threads=4, sims=3)):
        with pytest.raises(P.AzrError) as e:
            P.Engine(4, blocks=1, **({"sims": 8} | kw))
        assert e.value.code == 1   # AZR_E_INVALID_ARGUMENT


def test_create_destroy_does_not_leak_device_memory():
    P = pkg()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(6):
        eng = P.Engine(64, blocks=2, sims=8, dtype=P.NET_BF16 if i % 2 else P.NET_F32)
        eng.init_random(i)
        eng.selfplay_start(i)
        eng.selfplay_run(20)
        rec = eng.drain()
        if i % 3 == 0:   # the lazily created parts too: second tree + leaf lists of the two-net arena, optimiser context
            eng.arena_set_opponent(eng)
            eng.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B, 4, 0, True, 5)
            eng.arena_run(30)
            eng.arena_set_opponent(None)
            if len(rec) >= 32:
                eng.train_batch(rec[:32])
        eng.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_failed_create_hands_out_nothing_and_leaks_nothing():
    """azr_engine_create: on failure *out is NULL, azr_last_error(NULL) says why, nothing stays allocated — also when the
    failure comes after device allocations (an impossibly large record ring)."""
    P = pkg()
    L = P.load_library()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    # a node pool the 16-bit node index cannot address is rejected, not clamped (explicit, and through the default rule)
    for kw in (dict(node_capacity=65535), dict(sims=4095), dict(node_capacity=1 << 20)):
        with pytest.raises(P.AzrError) as e:
            P.Engine(4, blocks=1, **({"sims": 8} | kw))
        assert e.value.code == 1 and "65534" in str(e.value)
    ok = P.Engine(2, blocks=1, sims=8, node_capacity=65534)   # the largest pool is fine
    ok.close()
    # out of device memory half-way through: 4096 games x 2^20 records x 265 B cannot be allocated
    s = P.binding.Settings()
    L.azr_default_settings(C.byref(s))
    s.games, s.blocks, s.mcts_simulations, s.sample_capacity = 4096, 1, 8, 1 << 20
    h = C.c_void_p(0xdead)
    rc = L.azr_engine_create(C.byref(s), C.byref(h))
    assert rc == 4 and not h.value            # AZR_E_HIP, *out == NULL
    assert b"out of memory" in L.azr_last_error(None)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20


def test_partial_drain_keeps_the_rest_and_drops_are_counted(tmp_path):
    """azr_samples_drain with a small buffer returns the FIRST records and keeps the others; records_dropped counts
    what a too-small sample_capacity loses; a truncated checkpoint leaves the weights untouched."""
    P = pkg()
    eng = P.Engine(32, blocks=1, sims=4, dtype=P.NET_BF16, max_game_rounds=12)
    eng.init_random(2)
    eng.selfplay_start(9)
    while eng.counters()["games_finished"] < 8:
        eng.selfplay_run(64)
    c = eng.counters()
    assert c["records_dropped"] == 0
    n = eng.samples_device_view()[1]
    assert n == c["samples"] and n > 40
    import importlib
    full = importlib.import_module("alphazero-risk_amd.shard").device_records_to_torch(eng, torch.device("cuda", 0)).cpu().numpy()
    a = eng.drain(17)
    assert len(a) == 17 and eng.samples_device_view()[1] == n - 17
    b = eng.drain(5)
    rest = eng.drain()
    assert len(rest) == n - 22 and eng.samples_device_view()[1] == 0
    assert (np.concatenate([a, b, rest]) == full).all()     # nothing lost, nothing reordered
    eng.selfplay_run(64)
    eng.discard_samples()
    assert eng.samples_device_view()[1] == 0
    # truncated checkpoint: refused, weights unchanged
    w = eng.get_weights()
    eng.save(str(tmp_path / "ok.bin"))
    blob = (tmp_path / "ok.bin").read_bytes()
    (tmp_path / "short.bin").write_bytes(blob[:len(blob) // 2])
    (tmp_path / "long.bin").write_bytes(blob + b"\0\0\0\0")
    for name in ("short.bin", "long.bin"):
        with pytest.raises(P.AzrError) as e:
            eng.load(str(tmp_path / name))
        assert e.value.code == 6
        assert (eng.get_weights() == w).all()
    eng.close()
    # a game that outgrows sample_capacity loses records, and says so
    tiny = P.Engine(8, blocks=1, sims=4, dtype=P.NET_BF16, max_game_rounds=12, sample_capacity=8)
    tiny.init_random(2)
    tiny.selfplay_start(9)
    while tiny.counters()["games_finished"] < 4:
        tiny.selfplay_run(64)
    assert tiny.counters()["records_dropped"] > 0
    tiny.close()


def test_selfplay_quota_plays_exactly_the_asked_games(orc):
    """azr_selfplay_start_games (Counter::hasNext over TRAIN_ITERATION_GAMES, alphazero_trainer.cpp:83): exactly N games
    are started — seeds base .. base + N - 1 — and all are played to the end, whatever slot takes which; the record set
    equals the oracle's games of those seeds (device net called back)."""
    import azr_testlib as T
    P = pkg()
    G, sims, N, base = 6, 6, 15, 777
    eng = P.Engine(G, blocks=1, sims=sims, dtype=P.NET_F32, max_game_rounds=30, threads=1)
    eng.set_weights(T.make_net_flat(1, seed=11, perturb_bn=True))
    eng.selfplay_start_games(base, N)
    for _ in range(4000):
        eng.selfplay_run(64)
        c = eng.counters()
        if c["games_finished"] + c["errors"] >= N:
            break
    eng.selfplay_run(64)   # idle slots stay idle
    c = eng.counters()
    assert c["games_finished"] == N and c["errors"] == 0 and c["records_dropped"] == 0
    recs = eng.drain()
    assert len(recs) == c["samples"]

    @T.EVAL_FN
    def hip_eval(ctx, in88, pi, v):
        x = np.ctypeslib.as_array(in88, shape=(88,)).copy()[None]
        p, vv = eng.predict(x)
        C.memmove(pi, p.ctypes.data, 43 * 4)
        v[0] = float(vv[0])

    cfg = T.default_settings(mcts_simulations=sims, max_game_rounds=30, mcts_threads=1)
    blob, total = recs.tobytes(), 0
    for g in range(N):
        buf = np.zeros((4096, 265), np.uint8)
        st, rounds = C.c_int(0), C.c_int(0)
        n = orc.orc_selfplay_game(C.byref(cfg), base + g, hip_eval, None, T.ptr(buf), 4096, C.byref(st), C.byref(rounds),
                                  None, 0, None, None)
        assert n > 0 and buf[:n].tobytes() in blob, g
        total += n
    assert total == len(recs)
    # fewer games than slots
    eng.selfplay_start_games(base, 2)
    for _ in range(4000):
        eng.selfplay_run(64)
        if eng.counters()["games_finished"] >= 2:
            break
    eng.selfplay_run(64)
    assert eng.counters()["games_finished"] == 2
    eng.close()


def test_device_record_copy_equals_drain():
    """the device-side piece of the N > 1 path (trainStorage.extend per GPU, alphazero_trainer.cpp:59-62): the finished
    records copied device-to-device into a torch tensor (the collective's send buffer) are the bytes azr_samples_drain
    returns."""
    import importlib
    shard = importlib.import_module("alphazero-risk_amd.shard")
    P = pkg()
    eng = P.Engine(48, blocks=1, sims=6, dtype=P.NET_BF16, max_game_rounds=14)
    eng.init_random(4)
    eng.selfplay_start(31)
    while eng.counters()["games_finished"] < 6:
        eng.selfplay_run(64)
    n = eng.samples_device_view()[1]
    assert n > 0
    t = shard.device_records_to_torch(eng, torch.device("cuda", 0))
    assert t.shape == (n, 265) and t.dtype == torch.uint8 and t.is_cuda
    g = shard.gather_records(t, None)
    host = eng.drain()
    assert (g.cpu().numpy() == host).all()
    eng.close()
