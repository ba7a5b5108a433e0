#!/usr/bin/env python3
"""Diagnostic: average net launch time (HIP events on the engine stream, azr_profile_last_run) of self-play passes with
G games x T leaves, for the tile plans selected by AZR_TOWER_SB (read at engine creation).
    python tools/tower_time.py G T [passes] [modes...]        e.g.  tools/tower_time.py 256 2 300 1 3"""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]
G, T = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 300
modes = sys.argv[4:] or ["1"]
for rep in range(2):
    for mode in modes:
        os.environ["AZR_TOWER_SB"] = mode
        e = P.Engine(G, blocks=20, sims=100, dtype=P.NET_BF16, threads=T, test_hooks=not os.environ.get("AZR_EXP_LIB"))   # AZR_TOWER_SB is a test hook
        e.init_random(1)
        e.selfplay_start(1)
        e.selfplay_run(60)
        e.selfplay_run(passes)
        pr = e.profile_last_run()
        print(f"G {G} T {T} AZR_TOWER_SB={mode}: net {pr['net_ms'] * 1e3:.1f} us  tree {pr['tree_ms'] * 1e3:.1f} us over {pr['launches']} launches", flush=True)
        e.discard_samples()
        e.close()
