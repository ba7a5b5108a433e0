#!/usr/bin/env python3
"""Generates tests/golden/samples_io.npz from the REAL reference's sample storage (NNTrainDataStorage, the TF-free unit
neural_network/alphazero_nn_data.cpp:67-138,158-167, compiled in place into oracle/_ref/libazr_ref.so).

Run in the build container only:   python tests/golden/make_samples_golden.py

  records        [37,265]  the records handed to the reference (positions of encode.npz, seeded pi / z)
  file_by_ref    bytes     what saveTrainingSamples wrote for them: size_t count (8 B) + 265 B per record
  ref_reads_own  count + records the reference's OWN loadTrainingSamples gets back from that file: it consumes a 4-byte
                 count, so every record is read 4 bytes early (the header quirk, :92-93 vs :123-124)
  ref_reads_i32  the same reader on a file with the 4-byte count it expects: the records come back intact
  trim_*         trimOldExamples traces: (n, oldGameIndex, SAMPLES_STORAGE_MIN, SAMPLES_STORAGE_MAX) -> (n', first kept, oldGameIndex')
  extend_*       extend(a, b) + updateOldGamesIndex
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import azr_testlib as T  # noqa: E402


def main():
    R = T.ref()
    R.ref_save_samples.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    R.ref_load_samples.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    R.ref_trim_old_examples.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    R.ref_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(20260003)
    in88 = np.load(os.path.join(HERE, "encode.npz"))["in88"]
    n = 37
    rec = np.zeros((n, 265), np.uint8)
    rec[:, 0] = rng.integers(0, 2, n)
    rec[:, 1:89] = in88[np.linspace(0, len(in88) - 1, n).astype(int)]
    rec[:, 89:93] = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), n).view(np.uint8).reshape(n, 4)
    pi = rng.random((n, 43)).astype(np.float32)
    pi /= pi.sum(1, keepdims=True)
    rec[:, 93:265] = pi.view(np.uint8).reshape(n, 172)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "data", "training_samples.bin")
        R.ref_save_samples(T.ptr(rec), n, path.encode())
        file_by_ref = np.fromfile(path, np.uint8)
        own = np.zeros((n + 8, 265), np.uint8)
        own_n = R.ref_load_samples(path.encode(), T.ptr(own), len(own))
        p4 = os.path.join(d, "i32.bin")
        with open(p4, "wb") as f:
            f.write(np.int32(n).tobytes())
            f.write(rec.tobytes())
        i32 = np.zeros((n + 8, 265), np.uint8)
        i32_n = R.ref_load_samples(p4.encode(), T.ptr(i32), len(i32))
    cases = []
    for (nn, old, smin, smax) in [(100, 50, 200, 1000), (100, 30, 60, 1000), (100, 50, 90, 1000), (100, 10, 10, 40), (100, 0, 10, 1000),
                                  (1000, 999, 1, 1000), (1001, 5, 1, 1000), (64, 64, 32, 64), (65, 10, 32, 64), (5, 3, 5, 5), (6, 3, 5, 9)]:
        first, oo = C.c_int(0), C.c_long(0)
        n2 = R.ref_trim_old_examples(nn, old, smin, smax, C.byref(first), C.byref(oo))
        cases.append((nn, old, smin, smax, n2, first.value, oo.value))
    a, b = rec[:11].copy(), rec[20:29].copy()
    ext = np.zeros((40, 265), np.uint8)
    oi = C.c_long(0)
    ne = R.ref_extend(T.ptr(a), len(a), T.ptr(b), len(b), T.ptr(ext), len(ext), C.byref(oi))
    np.savez_compressed(os.path.join(HERE, "samples_io.npz"), records=rec, file_by_ref=file_by_ref, ref_reads_own_count=np.int64(own_n),
                        ref_reads_own=own[:min(own_n, len(own))], ref_reads_i32_count=np.int64(i32_n), ref_reads_i32=i32[:i32_n],
                        trim_cases=np.array(cases, np.int64), extend_a=a, extend_b=b, extend_out=ext[:ne], extend_old_index=np.int64(oi.value))
    print("file bytes", len(file_by_ref), "own reader count", own_n, "i32 reader count", i32_n, "trim", cases[:4], "extend", ne, oi.value)


if __name__ == "__main__":
    main()
