#!/usr/bin/env python3
"""Diagnostic: sustained in-kernel shader clock and launch time of the tower kernel for n boards (azr_debug_tower_clock:
s_memtime / s_memrealtime stamps of workgroup 0 after back-to-back warm-up launches on leaf buffers).
    python tools/tower_clock.py [n ...]        AZR_TOWER_SB=0|1|2 selects the tile plan (read at engine creation)"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]
L = P.load_library(test_hooks=not os.environ.get("AZR_EXP_LIB"))   # AZR_TOWER_SB is a test hook: the engines below live in the same library
L.azr_debug_tower_clock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sizes = [int(x) for x in sys.argv[1:]] or [256, 512, 768, 1024, 2048, 4096]
for n in sizes:
    e = P.Engine(n, blocks=20, sims=100, dtype=P.NET_BF16, threads=1, test_hooks=not os.environ.get("AZR_EXP_LIB"))   # AZR_TOWER_SB is a test hook
    e.init_random(1)
    e.selfplay_start(1)
    e.selfplay_run(20)
    ghz, ms = C.c_double(), C.c_double()
    rc = L.azr_debug_tower_clock(e.h, n, max(200, 600000 // n), C.byref(ghz), C.byref(ms))
    print(f"n {n} rc {rc} AZR_TOWER_SB={os.environ.get('AZR_TOWER_SB', '1')}: sustained shader clock {ghz.value:.3f} GHz, workgroup-0 tower {ms.value:.3f} ms", flush=True)
    e.close()
