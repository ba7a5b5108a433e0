#!/usr/bin/env python3
"""Where a tree step spends its time: runs the self-play move loop (from the start position, then from the mid-game positions of
tests/golden/rules_games.npz) on a timing-experiment build of the library (csrc built with -DAZR_TREE_PROF, see tools/tree_prof.sh)
whose tree-step wave adds 100-MHz ticks to per-segment sums; the library prints the table to stderr when the engine is closed.
    AZR_EXP_LIB=alphazero-risk_amd/csrc/dbg/libazr_prof.so python tools/tree_prof.py [--games 512] [--sims 100] [--threads 2] [--steps 20]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]
import bench  # noqa: E402  (midgame_states)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    passes = a.steps * (a.sims // a.threads + 1)
    for mid in (False, True):
        eng = P.Engine(a.games, blocks=a.blocks, sims=a.sims, dtype=P.NET_BF16, threads=a.threads)
        eng.init_random(20260002)
        if mid:
            st, seeds, _ = bench.midgame_states(a.games, 0)
            eng.set_states(st)
            eng.set_rng(seeds)
            eng.selfplay_start_from_states(20260001)
        else:
            eng.selfplay_start(20260001)
        eng.selfplay_run(5 * (a.sims // a.threads + 1))
        c0 = eng.counters()
        t0 = time.perf_counter()
        eng.selfplay_run(passes)
        dt = time.perf_counter() - t0
        c1 = eng.counters()
        pr = eng.profile_last_run()
        print(f"{'mid-game' if mid else 'start'}: {(c1['simulations'] - c0['simulations']) / dt:.0f} sims/s, tree step {pr['tree_ms']:.4f} ms, net {pr['net_ms']:.4f} ms, "
              f"depth {(c1['levels'] - c0['levels']) / max(1, c1['simulations'] - c0['simulations']):.2f}", flush=True)
        sys.stderr.write(f"[{'mid-game' if mid else 'start'}] ")
        sys.stderr.flush()
        eng.close()


if __name__ == "__main__":
    main()
