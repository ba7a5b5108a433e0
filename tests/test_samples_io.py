"""CPU: on-disk sample format and replay-buffer policy against what the REAL reference's storage did.

tests/golden/samples_io.npz was produced by tests/golden/make_samples_golden.py from the reference's own
NNTrainDataStorage (neural_network/alphazero_nn_data.cpp:67-138,158-167 — a TensorFlow-free unit compiled in place into
oracle/_ref).  Checked here for BOTH host implementations above the C-ABI — the C++ one (host/azr_host.cpp, through
tests/helpers/samples_probe.cpp) and the Python one (alphazero-risk_amd/learn.py):
  * the writer's bytes equal the reference writer's bytes (8-byte size_t count + 265 B per record)
  * the reader returns the records intact from that file AND from a file with the 4-byte count the reference's own
    reader expects; the reference's reader, given its own writer's file, returns records shifted by 4 bytes (the golden
    holds that too — the quirk is documented, not reproduced)
  * trimOldExamples / extend / updateOldGamesIndex traces
Where oracle/_ref is present (build container) the golden is re-derived live from the reference and compared."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import ROOT

G = np.load(os.path.join(T.GOLDEN, "samples_io.npz"))
HOST = os.path.join(ROOT, "alphazero-risk_amd", "host")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "alphazero-risk_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = str(tmp_path_factory.mktemp("probe") / "samples_probe")
    csrc = os.path.join(ROOT, "alphazero-risk_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", HOST, os.path.join(ROOT, "tests", "helpers", "samples_probe.cpp"),
                           os.path.join(HOST, "azr_host.o"), "-o", exe, "-L", csrc, "-lazr_hip", "-Wl,-rpath," + csrc])
    return exe


def learn():
    return importlib.import_module("alphazero-risk_amd.learn")


def test_golden_file_has_the_documented_layout():
    rec, raw = G["records"], G["file_by_ref"]
    assert len(raw) == 8 + len(rec) * 265
    assert int(raw[:8].view(np.uint64)[0]) == len(rec)
    assert (raw[8:].reshape(-1, 265) == rec).all()
    # the reference's reader on its own writer's file: count = low half of the size_t, every record read 4 bytes early
    assert int(G["ref_reads_own_count"]) == len(rec)
    shifted = raw[4:4 + len(rec) * 265].reshape(-1, 265)
    assert (G["ref_reads_own"] == shifted).all() and not (G["ref_reads_own"] == rec).all()
    # ... and on a file with the 4-byte count it expects: intact
    assert int(G["ref_reads_i32_count"]) == len(rec) and (G["ref_reads_i32"] == rec).all()


def test_python_writer_reader_trim(tmp_path):
    L = learn()
    rec = G["records"]
    p = str(tmp_path / "data" / "training_samples.bin")
    assert L.save_training_samples(p, rec)
    assert (np.fromfile(p, np.uint8) == G["file_by_ref"]).all()
    assert (L.load_training_samples(p) == rec).all()
    p4 = str(tmp_path / "i32.bin")
    with open(p4, "wb") as f:
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
    assert (L.load_training_samples(p4) == rec).all()
    assert len(L.load_training_samples(str(tmp_path / "missing.bin"))) == 0
    assert not L.save_training_samples(str(tmp_path / "empty.bin"), rec[:0]) and not os.path.exists(tmp_path / "empty.bin")
    with open(tmp_path / "junk.bin", "wb") as f:
        f.write(b"x" * 100)
    with pytest.raises(ValueError):
        L.load_training_samples(str(tmp_path / "junk.bin"))
    for n, old, smin, smax, n2, first, old2 in G["trim_cases"]:
        marks = np.arange(n)[:, None]
        r, o = L.trim_old_examples(marks, int(old), int(smin), int(smax))
        assert (len(r), int(r[0, 0]) if len(r) else -1, o) == (n2, first, old2), (n, old, smin, smax)
    # extend = concatenation in GPU order; updateOldGamesIndex = size - 1 (alphazero_nn_data.cpp:158-167)
    ext = np.concatenate([G["extend_a"], G["extend_b"]])
    assert (ext == G["extend_out"]).all() and max(len(ext) - 1, 0) == int(G["extend_old_index"])


def test_cpp_host_writer_reader_trim(probe, tmp_path):
    rec = G["records"]
    rec.tofile(tmp_path / "rec.bin")
    out = str(tmp_path / "data" / "training_samples.bin")
    subprocess.check_call([probe, "save", str(tmp_path / "rec.bin"), str(len(rec)), out])
    assert (np.fromfile(out, np.uint8) == G["file_by_ref"]).all()
    r = subprocess.run([probe, "load", out, str(tmp_path / "back.bin")], capture_output=True, text=True, check=True)
    assert f"count {len(rec)}" in r.stdout and (np.fromfile(tmp_path / "back.bin", np.uint8).reshape(-1, 265) == rec).all()
    with open(tmp_path / "i32.bin", "wb") as f:
        f.write(np.int32(len(rec)).tobytes()); f.write(rec.tobytes())
    r = subprocess.run([probe, "load", str(tmp_path / "i32.bin"), str(tmp_path / "back4.bin")], capture_output=True, text=True, check=True)
    assert f"count {len(rec)}" in r.stdout and (np.fromfile(tmp_path / "back4.bin", np.uint8).reshape(-1, 265) == rec).all()
    for n, old, smin, smax, n2, first, old2 in G["trim_cases"]:
        r = subprocess.run([probe, "trim", str(n), str(old), str(smin), str(smax)], capture_output=True, text=True, check=True)
        got = [int(x) for x in r.stdout.split("result")[1].split()]
        assert got == [n2, first, old2], (n, old, smin, smax, got)
    G["extend_a"].tofile(tmp_path / "a.bin"); G["extend_b"].tofile(tmp_path / "b.bin")
    r = subprocess.run([probe, "extend", str(tmp_path / "a.bin"), str(len(G["extend_a"])), str(tmp_path / "b.bin"), str(len(G["extend_b"])),
                        str(tmp_path / "ext.bin")], capture_output=True, text=True, check=True)
    assert [int(x) for x in r.stdout.split("result")[1].split()] == [len(G["extend_out"]), int(G["extend_old_index"])]
    assert (np.fromfile(tmp_path / "ext.bin", np.uint8).reshape(-1, 265) == G["extend_out"]).all()


@pytest.mark.ref
@pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref (the reference compiled in the build container) is absent")
def test_golden_is_what_the_reference_does_now(tmp_path):
    R = T.ref()
    R.ref_save_samples.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    R.ref_trim_old_examples.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rec = G["records"].copy()
    p = str(tmp_path / "d" / "s.bin")
    R.ref_save_samples(T.ptr(rec), len(rec), p.encode())
    assert (np.fromfile(p, np.uint8) == G["file_by_ref"]).all()
    for n, old, smin, smax, n2, first, old2 in G["trim_cases"]:
        f, o = C.c_int(0), C.c_long(0)
        assert R.ref_trim_old_examples(int(n), int(old), int(smin), int(smax), C.byref(f), C.byref(o)) == n2
        assert (f.value, o.value) == (first, old2)
