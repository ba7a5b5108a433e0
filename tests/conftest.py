import os
import sys

import pytest
import torch  # noqa: F401  first: torch bundles its own libamdhip64 (same soname as /opt/rocm's); loaded before the C-ABI
#              library, both bind to ONE HIP runtime (alphazero-risk_amd/shard.py:_one_hip_runtime)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled in the build container)")


@pytest.fixture(scope="session")
def orc():
    import azr_testlib as T
    return T.oracle()
