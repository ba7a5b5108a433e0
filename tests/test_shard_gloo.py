"""CPU, world_size 2 over gloo: the N>1 path of bench.py — disjoint per-rank seed streams, the padded all_gather of
(s, pi, z) records and the counter reduction (alphazero-risk_amd/shard.py)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    shard = importlib.import_module("alphazero-risk_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 0 if (world > 2 and rank == 5) else 5 + 7 * rank  # ragged counts, rank 0 smaller than rank 1; at world 8 one rank has nothing
    recs = torch.from_numpy(rng.integers(0, 256, (n, 265), dtype=np.uint8))
    allr = shard.gather_records(recs, dist)
    empty = shard.gather_records(torch.zeros((0, 265), dtype=torch.uint8) if rank == 0 else recs, dist)
    tot = shard.reduce_counters({"simulations": 10 * (rank + 1), "games_finished": rank}, dist)
    # the learn loop's weight hand-over: rank 0's vector reaches every rank
    flat = np.full(1000, float(rank + 1), np.float32) + np.arange(1000, dtype=np.float32)
    got = shard.broadcast_flat(flat, dist, src=0)
    assert (got == np.float32(1.0) + np.arange(1000, dtype=np.float32)).all()
    q.put((rank, recs.numpy(), allr.numpy(), empty.shape[0], tot, shard.rank_base_seed(20260001, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([res[0][1], res[1][1]])
    for r in range(world):
        assert (res[r][2] == want).all()          # every rank holds the rank-ordered concatenation
        assert res[r][3] == len(res[1][1])        # an empty shard contributes nothing
        assert res[r][4] == {"simulations": 30, "games_finished": 1}
    # seed ranges of different ranks cannot collide for < 2^24 games per slot-stream
    assert res[1][5] - res[0][5] == 1 << 24


def test_gather_records_world8_ragged_gloo():
    """configs[3] / configs[4] at their real rank count (on the CPU, gloo): 8 ragged record counts — one rank contributes none —
    gathered in rank order on every rank, counters summed, rank 0's weight vector broadcast, 8 disjoint seed ranges"""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    want = np.concatenate([res[r][1] for r in range(world)])
    assert len(want) == sum(5 + 7 * r for r in range(world) if r != 5) and len(res[5][1]) == 0
    for r in range(world):
        assert res[r][2].shape == want.shape and (res[r][2] == want).all()
        assert res[r][4] == {"simulations": 10 * 36, "games_finished": 28}
    seeds = [res[r][5] for r in range(world)]
    assert all(b - a == 1 << 24 for a, b in zip(seeds, seeds[1:]))


def test_split_count_covers_the_total():
    sys.path.insert(0, ROOT)
    shard = importlib.import_module("alphazero-risk_amd.shard")
    for total in (0, 1, 5, 50, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard.split_count(total, world, r) for r in range(world)]
            assert sum(parts) == total and max(parts) - min(parts) <= 1 and parts == sorted(parts, reverse=True)


def test_selfplay_seed_streams_never_collide():
    """the (iteration, rank) seed sets must be disjoint.  Every rank's stream is base + rank * 2^24 +
    (games it has started so far); with azr_selfplay_start_games game i of a batch plays seed + i."""
    import importlib
    shard = importlib.import_module("alphazero-risk_amd.shard")
    world = 8
    for tg in (1000, 256 * 8, 2048 * 8 + 5):
        started = [0] * world
        ranges = []
        for it in range(1000):
            for r in range(world):
                n = shard.split_count(tg, world, r)
                s0 = shard.selfplay_seed(20260001, r, started[r])
                started[r] += n
                ranges.append((s0, s0 + n))
                if started[r] + n >= shard.SEED_STRIDE:
                    break
        ranges.sort()
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 <= b0, (tg, (a0, a1), (b0, b1))
        # minstd_rand0 seeding folds seeds mod 2^31 - 1: the ranges stay below it
        assert ranges[-1][1] < 2**31 - 1
    import pytest
    with pytest.raises(ValueError):
        shard.selfplay_seed(1, 0, shard.SEED_STRIDE)
