"""PUCT tie-break order = libstdc++ unordered_map iteration order.  The oracle's
emulation is validated against the real host library (g++ is in the image, here and on the GPU box)."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import azr_testlib as T


def masks():
    rng = np.random.default_rng(5)
    m = [(1 << 43) - 1, 1, 1 << 42, 0b11, (1 << 13) - 1, (1 << 14) - 1, (1 << 29) - 1, (1 << 30) - 1,
         (1 << 42) | (1 << 13) | 1, sum(1 << i for i in range(0, 43, 13))]
    for n in range(1, 44):
        for _ in range(12):
            m.append(int(sum(1 << int(i) for i in rng.choice(43, n, replace=False))))
    return m


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_emulation_matches_host_libstdcxx(orc):
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "umap_probe")
    subprocess.check_call(["g++", "-std=gnu++2a", "-O1", "-o", exe,
                           os.path.join(T.ROOT, "tests", "helpers", "umap_probe.cpp")])
    ms = masks()
    out = subprocess.run([exe], input="\n".join("%x" % m for m in ms), capture_output=True, text=True, check=True)
    lines = out.stdout.strip().split("\n")
    assert len(lines) == len(ms)
    buf = (C.c_uint8 * 43)()
    for m, line in zip(ms, lines):
        n = orc.orc_umap_order(m, buf)
        assert [buf[i] for i in range(n)] == [int(x) for x in line.split()], hex(m)


def test_full_mask_order_matches_survey_probe(orc):
    buf = (C.c_uint8 * 43)()
    n = orc.orc_umap_order((1 << 43) - 1, buf)
    assert [buf[i] for i in range(n)] == list(range(42, 28, -1)) + list(range(12, -1, -1)) + list(range(13, 29))
