"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/azr.h declares;
the product refuses to run (no fallback) when there is no device."""
import os
import re

import pytest

from gpu_common import pkg, ROOT


def test_builds_loads_and_exports_every_declared_symbol():
    P = pkg()
    P.build()
    L = P.load_library()
    hdr = open(os.path.join(ROOT, "include", "azr.h")).read()
    declared = set(re.findall(r"\b(azr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert set(P.binding.EXPORTS) == declared


def test_param_count_matches_reference_shapes():
    L = pkg().load_library()
    F = 256
    for blocks, expect in ((5, None), (20, None)):
        n = 9 * 13 * F + 28 + blocks * 2 * (9 * F * F + 4 * F) + (F * 2 + 8 + 84 * 43 + 43) + (F + 4 + 42 * 256 + 256 + 256 + 1)
        assert L.azr_nn_param_count(blocks) == n
    # ~23.66 M parameters at B = 20 (BN moving stats included here)
    assert 23.6e6 < L.azr_nn_param_count(20) < 23.8e6


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    P = pkg()
    with pytest.raises(P.AzrError):
        P.Engine(2, blocks=1, sims=1)


def test_header_constants_match_the_binding():
    """the enumerators of include/azr.h that the Python side restates (net arithmetic, player kinds) carry the header's values"""
    P = pkg()
    hdr = open(os.path.join(ROOT, "include", "azr.h")).read()
    vals = {k: int(v) for k, v in re.findall(r"\b(AZR_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    for name in ("F32", "BF16", "F32X", "F16"):
        assert vals["AZR_NET_" + name] == getattr(P, "NET_" + name), name
    for name in ("ALPHAZERO", "SCRIPT", "RANDOM", "ALPHAZERO_B"):
        assert vals["AZR_PLAYER_" + name] == getattr(P, "PLAYER_" + name), name
    for name in ("OFF", "SEQUENTIAL", "CONCURRENT"):
        assert vals["AZR_MIRROR_" + name] == getattr(P, "MIRROR_" + name), name
