#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of one k_tower_sb4 launch (azr_debug_tower_trace) after warm-up launches.
    python tools/tower_trace.py [n ...]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
P = importlib.import_module("alphazero-risk_amd")
if os.environ.get("AZR_EXP_LIB"):   # a timing-experiment build of the same sources (never the product library)
    P.binding.lib_path = lambda test_hooks=False: os.environ["AZR_EXP_LIB"]
L = P.load_library()
L.azr_debug_tower_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
for n in [int(x) for x in sys.argv[1:]] or [1024, 4096]:
    e = P.Engine(n, blocks=20, sims=100, dtype=P.NET_BF16, threads=1)
    e.init_random(1)
    e.selfplay_start(1)
    e.selfplay_run(20)
    out = np.zeros((n, 8), np.uint64)
    wgs = C.c_int(0)
    rc = L.azr_debug_tower_trace(e.h, n, 300, out.ctypes.data_as(C.c_void_p), n, C.byref(wgs))
    t = out[:wgs.value].astype(np.int64)
    if rc or not len(t):
        print(n, "rc", rc, "no trace (kernel without per-workgroup stamps)")
        continue
    t0 = t[:, 0].min()
    us = lambda x: x / 100.0   # 100 MHz ticks -> microseconds
    life, tower, setup, heads = t[:, 3] - t[:, 0], t[:, 2] - t[:, 1], t[:, 1] - t[:, 0], t[:, 3] - t[:, 2]
    print(f"n {n}: {len(t)} workgroups; launch span {us(t[:, 3].max() - t0):.1f} us; start skew {us(t[:, 0].max() - t0):.1f} us")
    for name, a in (("lifetime", life), ("setup+stem", setup), ("tower", tower), ("heads", heads)):
        print(f"   {name:11s} min {us(a.min()):8.1f}  p50 {us(np.median(a)):8.1f}  max {us(a.max()):8.1f} us")
    for x in range(8):
        m = t[:, 4] == x
        if m.any():
            ghz = np.median((t[m, 6] - t[m, 5]) / np.maximum(tower[m], 1)) * 0.1
            print(f"   XCC {x}: {m.sum():4d} workgroups, tower p50 {us(np.median(tower[m])):8.1f} max {us(tower[m].max()):8.1f} us at {ghz:.3f} GHz = "
                  f"{np.median(t[m, 6] - t[m, 5]) / 1e6:.3f} Mcycles; first start {us(t[m, 0].min() - t0):6.1f} last end {us(t[m, 3].max() - t0):8.1f}")
    e.close()
