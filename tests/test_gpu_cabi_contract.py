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
