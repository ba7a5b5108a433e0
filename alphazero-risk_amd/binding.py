"""ctypes binding of include/azr.h (libazr_hip.so).  No compute happens in Python."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
NET_F32, NET_BF16, NET_F32X, NET_F16 = 0, 1, 2, 3
MOVES, STATE_BYTES, INPUT_BYTES, RECORD_BYTES = 43, 160, 88, 265


class AzrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"azr error {code}: {msg}")
        self.code = code


class Settings(C.Structure):
    """`azr_settings` — mirrors the reference's `class Settings` fields the hot path reads (src/settings.h:41-64)."""
    _fields_ = [("device", C.c_int32), ("games", C.c_int32), ("blocks", C.c_int32), ("net_dtype", C.c_int32),
                ("mcts_simulations", C.c_int32), ("mcts_threads", C.c_int32), ("allow_yield", C.c_int32), ("limit_reinforcement", C.c_int32),
                ("limit_attack", C.c_int32), ("max_game_rounds", C.c_int32), ("min_unit_move", C.c_int32),
                ("temperature_threshold", C.c_int32), ("hp_exploration", C.c_float), ("dir_noise_value", C.c_float),
                ("dir_noise_epsi", C.c_float), ("node_capacity", C.c_int32), ("sample_capacity", C.c_int32)]


class GameResults(C.Structure):
    """`azr_game_results` = GameResults (game/game.h:17-29)"""
    _fields_ = [("count", C.c_int32), ("draw", C.c_int32), ("win", C.c_int32 * 2), ("win_and_started", C.c_int32 * 2)]

    def as_dict(self):
        return dict(count=self.count, draw=self.draw, win=list(self.win), win_and_started=list(self.win_and_started))


PLAYER_ALPHAZERO, PLAYER_SCRIPT, PLAYER_RANDOM, PLAYER_ALPHAZERO_B = 0, 1, 2, 3
MIRROR_OFF, MIRROR_SEQUENTIAL, MIRROR_CONCURRENT = 0, 1, 2   # azr_arena_start's mirror_games (include/azr.h)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("simulations", "evaluations", "levels", "decisions", "games_finished",
                                          "samples", "nodes_dropped", "errors", "records_dropped", "tower_fallbacks")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)   # azr_allreduce_fn


def dp_unique_id():
    """azr_dp_unique_id: the 128-byte id rank 0 draws and hands to the other ranks"""
    b = (C.c_uint8 * 128)()
    rc = load_library().azr_dp_unique_id(b)
    if rc:
        raise AzrError(rc, "azr_dp_unique_id: RCCL is not available")
    return bytes(b)


def lib_path(test_hooks=False):
    """the product library, or (test_hooks) libazr_hip_test.so: the same sources compiled with -DAZR_TEST_HOOKS — the only build
    in which the AZR_TOWER_* / AZR_TRAIN_* / AZR_DP_LOOPBACK environment switches exist (csrc/azr_internal.hpp)"""
    return os.path.join(CSRC, "libazr_hip_test.so" if test_hooks else "libazr_hip.so")


def build(jobs=4, test_hooks=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", CSRC, "test" if test_hooks else "all"])
    return lib_path(test_hooks)


_lib = None
_libs = {}

EXPORTS = [
    "azr_default_settings", "azr_engine_create", "azr_engine_destroy", "azr_last_error", "azr_engine_games",
    "azr_engine_new_games", "azr_engine_set_states", "azr_engine_get_states", "azr_engine_set_rng", "azr_engine_get_rng",
    "azr_engine_valid_moves", "azr_engine_make_moves", "azr_engine_status", "azr_engine_encode",
    "azr_nn_param_count", "azr_nn_init_random", "azr_nn_set_weights", "azr_nn_get_weights", "azr_nn_load", "azr_nn_save",
    "azr_nn_predict", "azr_nn_train", "azr_nn_train_dp", "azr_dp_unique_id", "azr_dp_init", "azr_dp_shutdown", "azr_nn_train_batch", "azr_nn_train_grads", "azr_nn_train_reset", "azr_mcts_clear", "azr_mcts_trim", "azr_mcts_simulate", "azr_mcts_begin", "azr_mcts_leaves",
    "azr_mcts_apply", "azr_mcts_root_stats", "azr_mcts_policy", "azr_mcts_pick", "azr_selfplay_start", "azr_selfplay_start_games", "azr_selfplay_start_from_states",
    "azr_selfplay_run", "azr_selfplay_counters", "azr_samples_drain", "azr_samples_device_view", "azr_samples_copy_device", "azr_profile_last_run",
    "azr_device_synchronize", "azr_debug_tower_clock", "azr_debug_tower_trace", "azr_debug_tower_plan", "azr_arena_start", "azr_arena_run", "azr_arena_results", "azr_arena_log",
    "azr_arena_set_opponent_net", "azr_arena_collect_samples",
]


def load_library(test_hooks=False):
    """dlopen libazr_hip.so (or the test-hook build); raises (never falls back) when it has not been built."""
    global _lib
    if test_hooks not in _libs:
        p = lib_path(test_hooks)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(the HIP path is the only path; there is no CPU fallback)")
        L = C.CDLL(p)
        L.azr_last_error.restype = C.c_char_p
        L.azr_last_error.argtypes = [C.c_void_p]
        L.azr_nn_param_count.restype = C.c_size_t
        L.azr_nn_param_count.argtypes = [C.c_int]
        L.azr_engine_create.argtypes = [C.c_void_p, C.c_void_p]
        for name in EXPORTS:
            f = getattr(L, name)
            if f.argtypes is None and name not in ("azr_default_settings",):
                pass
        L.azr_nn_init_random.argtypes = [C.c_void_p, C.c_uint64]
        L.azr_nn_set_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.azr_nn_get_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.azr_nn_load.argtypes = [C.c_void_p, C.c_char_p]
        L.azr_nn_save.argtypes = [C.c_void_p, C.c_char_p]
        L.azr_nn_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.azr_nn_train.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_nn_train_dp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_nn_train_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.azr_dp_unique_id.argtypes = [C.c_void_p]
        L.azr_dp_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.azr_dp_shutdown.argtypes = [C.c_void_p]
        L.azr_nn_train_grads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.azr_nn_train_reset.argtypes = [C.c_void_p]
        L.azr_selfplay_start.argtypes = [C.c_void_p, C.c_uint32]
        L.azr_selfplay_start_games.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.azr_selfplay_start_from_states.argtypes = [C.c_void_p, C.c_uint32]
        L.azr_samples_copy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.azr_selfplay_run.argtypes = [C.c_void_p, C.c_int]
        L.azr_samples_drain.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.azr_samples_device_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_mcts_pick.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.azr_mcts_leaves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_mcts_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_mcts_root_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_profile_last_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_debug_tower_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        for name in ("azr_engine_destroy", "azr_engine_games", "azr_mcts_clear", "azr_mcts_trim", "azr_mcts_simulate",
                     "azr_mcts_begin", "azr_device_synchronize"):
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("azr_engine_new_games", "azr_engine_set_states", "azr_engine_get_states", "azr_engine_set_rng",
                     "azr_engine_get_rng", "azr_engine_valid_moves", "azr_engine_status", "azr_engine_encode",
                     "azr_mcts_policy", "azr_selfplay_counters"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.azr_engine_make_moves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.azr_arena_start.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.azr_arena_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.azr_arena_set_opponent_net.argtypes = [C.c_void_p, C.c_void_p]
        L.azr_arena_collect_samples.argtypes = [C.c_void_p, C.c_int]
        L.azr_arena_results.argtypes = [C.c_void_p, C.c_void_p]
        L.azr_arena_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _libs[test_hooks] = L
        if not test_hooks:
            _lib = L
    return _libs[test_hooks]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One handle = one GPU = G concurrent games.  Method names follow the reference seams:
    State / UtilityNN (rules), AlphaZeroNNId (net), AlphaZeroMCTS (search), trainer move loop (self-play)."""

    def __init__(self, games, blocks=20, sims=32, dtype=NET_BF16, device=0, threads=1, test_hooks=False, **kw):
        """threads = THREADS_PER_MCTS.  The C default (azr_default_settings) is the reference's 2; this binding defaults
        to 1, the only value at which the reference's search is deterministic and the parity tests are bit-exact
        against `-t 1` semantics; tests of the T-thread schedule pass it explicitly.  test_hooks: the handle lives in
        libazr_hip_test.so (tests only: the build whose environment switches select other formulations)."""
        self.L = load_library(test_hooks)
        s = Settings()
        self.L.azr_default_settings(C.byref(s))
        s.device, s.games, s.blocks, s.mcts_simulations, s.net_dtype = device, games, blocks, sims, dtype
        s.mcts_threads = threads
        self.T = threads
        for k, v in kw.items():
            if not hasattr(s, k):
                raise TypeError(f"unknown setting {k}")
            setattr(s, k, v)
        self.settings = s
        self.G = games
        self.blocks = blocks
        self.h = C.c_void_p()
        rc = self.L.azr_engine_create(C.byref(s), C.byref(self.h))
        if rc:  # *out is NULL on failure; the reason is kept per thread
            self.h = None
            raise AzrError(rc, self.L.azr_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.azr_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise AzrError(rc, self.L.azr_last_error(self.h).decode())

    # ---- rules
    def new_games(self, seeds):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        assert seeds.shape == (self.G,)
        self._chk(self.L.azr_engine_new_games(self.h, _p(seeds)))

    def set_states(self, data160):
        d = np.ascontiguousarray(data160, np.uint8)
        assert d.shape == (self.G, 160)
        self._chk(self.L.azr_engine_set_states(self.h, _p(d)))

    def get_states(self):
        d = np.zeros((self.G, 160), np.uint8)
        self._chk(self.L.azr_engine_get_states(self.h, _p(d)))
        return d

    def set_rng(self, x):
        x = np.ascontiguousarray(x, np.uint32)
        assert x.shape == (self.G,)
        self._chk(self.L.azr_engine_set_rng(self.h, _p(x)))

    def get_rng(self):
        x = np.zeros(self.G, np.uint32)
        self._chk(self.L.azr_engine_get_rng(self.h, _p(x)))
        return x

    def valid_moves(self):
        m = np.zeros(self.G, np.uint64)
        self._chk(self.L.azr_engine_valid_moves(self.h, _p(m)))
        return m

    def make_moves(self, moves):
        mv = np.ascontiguousarray(moves, np.uint8)
        assert mv.shape == (self.G,)
        rc = np.zeros(self.G, np.uint8)
        self._chk(self.L.azr_engine_make_moves(self.h, _p(mv), _p(rc)))
        return rc

    def status(self):
        s = np.zeros(self.G, np.int8)
        self._chk(self.L.azr_engine_status(self.h, _p(s)))
        return s

    def encode(self):
        e = np.zeros((self.G, 88), np.uint8)
        self._chk(self.L.azr_engine_encode(self.h, _p(e)))
        return e

    # ---- net
    def param_count(self):
        return self.L.azr_nn_param_count(self.blocks)

    def init_random(self, seed=20260002):
        self._chk(self.L.azr_nn_init_random(self.h, seed))

    def set_weights(self, flat):
        f = np.ascontiguousarray(flat, np.float32)
        self._chk(self.L.azr_nn_set_weights(self.h, _p(f), f.size))

    def get_weights(self):
        f = np.zeros(self.param_count(), np.float32)
        self._chk(self.L.azr_nn_get_weights(self.h, _p(f), f.size))
        return f

    def save(self, path):
        self._chk(self.L.azr_nn_save(self.h, path.encode()))

    def load(self, path):
        self._chk(self.L.azr_nn_load(self.h, path.encode()))

    def predict(self, in88):
        x = np.ascontiguousarray(in88, np.uint8)
        n = x.shape[0]
        assert x.shape == (n, 88)
        pi = np.zeros((n, 43), np.float32)
        v = np.zeros(n, np.float32)
        self._chk(self.L.azr_nn_predict(self.h, _p(x), n, _p(pi), _p(v)))
        return pi, v

    def train(self, rec265, epochs, batch_size=512, rng_state=None):
        """AlphaZeroNNId::train: returns ([(loss_pi, loss_v)] per epoch, engine state after the shuffles)"""
        r = np.ascontiguousarray(rec265, np.uint8).reshape(-1, 265)
        lp = np.zeros(max(epochs, 1), np.float32)
        lv = np.zeros(max(epochs, 1), np.float32)
        st = np.array([rng_state if rng_state is not None else 1], np.uint32)
        self._chk(self.L.azr_nn_train(self.h, _p(r), len(r), epochs, batch_size, _p(st) if rng_state is not None else None,
                                      _p(lp), _p(lv)))
        return [(float(lp[e]), float(lv[e])) for e in range(epochs)], int(st[0])

    def dp_init(self, rank, world, id128):
        """azr_dp_init: this handle's RCCL communicator (collective over the `world` processes); id128 = dp_unique_id() of rank 0"""
        b = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self._chk(self.L.azr_dp_init(self.h, rank, world, b))

    def dp_shutdown(self):
        self._chk(self.L.azr_dp_shutdown(self.h))

    def train_dp(self, rec265, epochs, allreduce, rank, world, batch_size=512, rng_state=None):
        """azr_nn_train_dp: this rank's share of a data-parallel AlphaZeroNNId::train.  allreduce(device_ptr, count, dtype)
        sums a device buffer over the ranks in place (dtype 0 = float32, 1 = float64); see shard.make_allreduce.  allreduce = None:
        the handle's own RCCL communicator (dp_init), every sum in stream order."""
        if allreduce is None:
            r = np.ascontiguousarray(rec265, np.uint8).reshape(-1, 265)
            lp = np.zeros(max(epochs, 1), np.float32)
            lv = np.zeros(max(epochs, 1), np.float32)
            st = np.array([rng_state if rng_state is not None else 1], np.uint32)
            self._chk(self.L.azr_nn_train_dp(self.h, _p(r), len(r), epochs, batch_size, _p(st) if rng_state is not None else None, rank, world,
                                             C.cast(None, ALLREDUCE_FN), None, _p(lp), _p(lv)))
            return [(float(lp[e]), float(lv[e])) for e in range(epochs)], int(st[0])
        r = np.ascontiguousarray(rec265, np.uint8).reshape(-1, 265)
        lp = np.zeros(max(epochs, 1), np.float32)
        lv = np.zeros(max(epochs, 1), np.float32)
        st = np.array([rng_state if rng_state is not None else 1], np.uint32)

        def cb(ctx, ptr, count, dtype):
            try:
                allreduce(ptr, count, dtype)
                return 0
            except Exception as e:   # never let an exception cross the C frame
                self._dp_error = e
                return 1

        fn = ALLREDUCE_FN(cb)
        self._dp_error = None
        rc = self.L.azr_nn_train_dp(self.h, _p(r), len(r), epochs, batch_size, _p(st) if rng_state is not None else None, rank, world,
                                    fn, None, _p(lp), _p(lv))
        if rc and self._dp_error is not None:
            raise self._dp_error
        self._chk(rc)
        return [(float(lp[e]), float(lv[e])) for e in range(epochs)], int(st[0])

    def train_batch(self, rec265):
        """one optimiser step on exactly these records; returns (loss_pi, loss_v)"""
        r = np.ascontiguousarray(rec265, np.uint8).reshape(-1, 265)
        l = np.zeros(2, np.float32)
        self._chk(self.L.azr_nn_train_batch(self.h, _p(r), len(r), _p(l[0:1]), _p(l[1:2])))
        return float(l[0]), float(l[1])

    def train_grads(self):
        g = np.zeros(self.L.azr_nn_param_count(self.settings.blocks), np.float32)
        self._chk(self.L.azr_nn_train_grads(self.h, _p(g), g.size))
        return g

    def train_reset(self):
        self._chk(self.L.azr_nn_train_reset(self.h))

    # ---- search
    def mcts_clear(self):
        self._chk(self.L.azr_mcts_clear(self.h))

    def mcts_trim(self):
        self._chk(self.L.azr_mcts_trim(self.h))

    def simulate(self):
        self._chk(self.L.azr_mcts_simulate(self.h))

    def mcts_begin(self):
        self._chk(self.L.azr_mcts_begin(self.h))

    def mcts_leaves(self):
        x = np.zeros((self.G * self.T, 88), np.uint8)      # slot = g * T + k
        need = np.zeros(self.G * self.T, np.uint8)
        act = C.c_int(0)
        self._chk(self.L.azr_mcts_leaves(self.h, _p(x), _p(need), C.byref(act)))
        return x, need.astype(bool), act.value

    def mcts_apply(self, pi, v):
        pi = np.ascontiguousarray(pi, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        assert pi.shape == (self.G * self.T, 43) and v.shape == (self.G * self.T,)
        self._chk(self.L.azr_mcts_apply(self.h, _p(pi), _p(v)))

    def root_stats(self):
        n = np.zeros((self.G, 43), np.uint32)
        q = np.zeros((self.G, 43), np.float32)
        p = np.zeros((self.G, 43), np.float32)
        self._chk(self.L.azr_mcts_root_stats(self.h, _p(n), _p(q), _p(p)))
        return n, q, p

    def policy(self):
        pi = np.zeros((self.G, 43), np.float32)
        self._chk(self.L.azr_mcts_policy(self.h, _p(pi)))
        return pi

    def pick(self, sample=False):
        m = np.zeros(self.G, np.uint8)
        self._chk(self.L.azr_mcts_pick(self.h, int(sample), _p(m)))
        return m

    # ---- self-play
    def selfplay_start(self, base_seed=20260001):
        self._chk(self.L.azr_selfplay_start(self.h, base_seed))

    def selfplay_start_games(self, base_seed, games):
        """exactly `games` games (seeds base_seed .. base_seed + games - 1), each played to its end"""
        self._chk(self.L.azr_selfplay_start_games(self.h, base_seed, games))

    def selfplay_start_from_states(self, base_seed=20260001):
        """self-play goes on from the states / RNG streams set with set_states / set_rng"""
        self._chk(self.L.azr_selfplay_start_from_states(self.h, base_seed))

    def selfplay_run(self, passes):
        self._chk(self.L.azr_selfplay_run(self.h, passes))

    def counters(self):
        c = Counters()
        self._chk(self.L.azr_selfplay_counters(self.h, C.byref(c)))
        return c.as_dict()

    def drain(self, cap=None):
        """the first `cap` buffered records (default: all); what does not fit stays buffered"""
        if cap is None:
            cap = max(1, self.samples_device_view()[1])
        buf = np.empty((cap, RECORD_BYTES), np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.azr_samples_drain(self.h, _p(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    def discard_samples(self):
        self._chk(self.L.azr_samples_drain(self.h, None, 0, None))

    def samples_copy_device(self, dst_ptr, cap):
        """copy up to `cap` buffered records to device memory at dst_ptr (engine stream); returns the count"""
        n = C.c_size_t(0)
        self._chk(self.L.azr_samples_copy_device(self.h, C.c_void_p(dst_ptr), cap, C.byref(n)))
        return n.value

    def samples_device_view(self):
        ptr = C.c_void_p()
        n = C.c_size_t(0)
        self._chk(self.L.azr_samples_device_view(self.h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def tower_plan(self, n):
        """(boards per workgroup, workgroups) of a bf16 net launch of n boards"""
        nb, w = C.c_int(0), C.c_int(0)
        self._chk(self.L.azr_debug_tower_plan(self.h, n, C.byref(nb), C.byref(w)))
        return nb.value, w.value

    def profile_last_run(self):
        a, b, k = C.c_float(0), C.c_float(0), C.c_int(0)
        self._chk(self.L.azr_profile_last_run(self.h, C.byref(a), C.byref(b), C.byref(k)))
        return dict(net_ms=a.value, tree_ms=b.value, launches=k.value)

    # ---- arena (GameGroup::playGames)
    def arena_start(self, player1, player2, games, per_slot_cap=0, mirror=True, base_seed=20260001):
        """mirror: False / True (= MIRROR_SEQUENTIAL, the reference's thread-per-pair form) / MIRROR_CONCURRENT (a pair's two games at
        the same time on slots 2j, 2j + 1)"""
        self._chk(self.L.azr_arena_start(self.h, player1, player2, games, per_slot_cap, int(mirror), base_seed))

    def arena_set_opponent(self, other):
        """the network of PLAYER_ALPHAZERO_B = `other`'s (an Engine on the same device, or None to detach)"""
        self._chk(self.L.azr_arena_set_opponent_net(self.h, other.h if other is not None else None))
        self._opponent = other   # keep it alive

    def arena_collect_samples(self, on=True):
        self._chk(self.L.azr_arena_collect_samples(self.h, int(on)))

    def arena_run(self, passes):
        fin = C.c_int(0)
        self._chk(self.L.azr_arena_run(self.h, passes, C.byref(fin)))
        return bool(fin.value)

    def arena_results(self):
        r = GameResults()
        self._chk(self.L.azr_arena_results(self.h, C.byref(r)))
        return r.as_dict()

    def arena_log(self):
        n = np.zeros(self.G, np.int32)
        st = np.zeros((self.G, 16), np.int8)
        rd = np.zeros((self.G, 16), np.uint16)
        fin = np.zeros((self.G, 16, 160), np.uint8)
        self._chk(self.L.azr_arena_log(self.h, _p(n), _p(st), _p(rd), _p(fin)))
        return n, st, rd, fin

    def synchronize(self):
        self._chk(self.L.azr_device_synchronize(self.h))
