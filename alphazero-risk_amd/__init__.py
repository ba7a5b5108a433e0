"""alphazero-risk_amd — MI355X-native AlphaZero-Risk self-play hot path.

The product is the HIP shared library `csrc/libazr_hip.so` behind the C-ABI of `include/azr.h`;
this package is the thin ctypes binding used by tests, bench.py and the Python-side tooling.
It fails loudly when the HIP library is missing (there is no CPU fallback).
"""
from .binding import Engine, Settings, Counters, build, lib_path, load_library, AzrError, dp_unique_id  # noqa: F401
from .binding import PLAYER_ALPHAZERO, PLAYER_SCRIPT, PLAYER_RANDOM, PLAYER_ALPHAZERO_B, GameResults  # noqa: F401
from .binding import MIRROR_OFF, MIRROR_SEQUENTIAL, MIRROR_CONCURRENT  # noqa: F401
from .binding import NET_F32, NET_BF16, NET_F32X, NET_F16, MOVES, STATE_BYTES, INPUT_BYTES, RECORD_BYTES  # noqa: F401
