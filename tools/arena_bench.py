#!/usr/bin/env python3
"""Time the new-vs-old arena of the learn loop (GameGroup::playGames with two AlphaZero players, two networks) on the
device: `games` mirrored games on `slots` engine slots, S simulations per move, THREADS_PER_MCTS T, B blocks, bf16.
    python tools/arena_bench.py [--games 100] [--slots 128] [--sims 100] [--threads 2] [--blocks 20]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=100)
    ap.add_argument("--slots", type=int, default=128)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--pair-halves", type=int, default=1, help="1: a mirrored pair's two games at the same time on two slots; 0: one after the other on one slot")
    a = ap.parse_args()
    new = P.Engine(a.slots, blocks=a.blocks, sims=a.sims, dtype=P.NET_BF16, threads=a.threads)
    old = P.Engine(a.slots, blocks=a.blocks, sims=a.sims, dtype=P.NET_BF16, threads=a.threads)
    new.init_random(1)
    old.init_random(2)
    new.arena_set_opponent(old)
    new.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B, a.games, 0, P.MIRROR_CONCURRENT if a.pair_halves else P.MIRROR_SEQUENTIAL, 20260001)
    t0 = time.time()
    while not new.arena_run(256):
        pass
    dt = time.time() - t0
    r = new.arena_results()
    print(f"{a.games} games on {a.slots} slots, S={a.sims} T={a.threads} B={a.blocks}: {dt:.1f} s, results {r}", flush=True)


if __name__ == "__main__":
    main()
