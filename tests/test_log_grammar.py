"""CPU: the three log files of the learn loop keep the reference's grammar, so python/src/log_chart.py
keeps reading them.  Fixtures: the authors' own log files (tests/golden/ref_logs.json, output data of their runs).  The grammar is
written once below, checked against every line the reference ever wrote, then against the lines the Python learn loop
(alphazero-risk_amd/learn.py) and the C++ host (host/azr_host.cpp through tests/helpers/samples_probe.cpp) format —
and, on the GPU box, against the files an end-to-end iteration leaves behind (tests/test_learn.py, test_host_cli.py)."""
import csv
import importlib
import io
import os
import re
import subprocess

import pytest

import azr_testlib as T
from gpu_common import ROOT

import json

REF = json.load(open(os.path.join(T.GOLDEN, "ref_logs.json")))   # the authors' own log files (tests/golden/make_logs_golden.py)


def ref_text(kind):
    return REF[kind]["text"]


def ref_first_line(kind):
    return REF[kind]["text"].split("\n")[0] + "\n"
PLAYER = r"\d+/\d+"
GR = rf"\d+, {PLAYER}, {PLAYER}"                       # operator<<(GameResults), game.cpp:227-235: draw, W/Wstart, W/Wstart
IMPROVEMENT = re.compile(rf"^\d+,{GR}$")                # alphazero_trainer.cpp:163
BENCHMARK = re.compile(rf"^\d+,{GR}, {GR}$")            # alphazero_trainer.cpp:139
FLOAT = r"-?(?:\d+\.?\d*|\.\d+)(?:e[-+]?\d+)?|nan|inf"
NN_TRAINING = re.compile(rf"^(?:(?:{FLOAT}), (?:{FLOAT}), ?)*$")  # "policy, value, " per epoch (some of the authors' lines lost the last blank)


def chart_parse(kind, text):
    """what python/src/log_chart.py does with a file (its csv.reader + int/split logic), returning the parsed rows"""
    rows = []
    for row in csv.reader(io.StringIO(text), delimiter=","):
        if kind == "improvement":      # GameResults(row)
            rows.append((int(row[0]), int(row[1])) + tuple(int(x) for x in row[2].split("/")) + tuple(int(x) for x in row[3].split("/")))
        elif kind == "benchmark":      # GameResults(row[0:4]), GameResults(row[0:1] + row[4:])
            assert len(row) == 7
            rows.append((int(row[0]), int(row[1]), int(row[4])) + tuple(int(x) for f in (row[2], row[3], row[5], row[6]) for x in f.split("/")))
        else:                          # float(row[0]), float(row[1]), ... ; the trailing ", " leaves one blank field
            assert not row or row[-1].strip() == ""      # (an empty line = a train() call with fewer records than one batch)
            rows.append(tuple(float(x) for x in row[:-1]))
    return rows


def check(kind, text, complete=True):
    rx = {"improvement": IMPROVEMENT, "benchmark": BENCHMARK, "nn": NN_TRAINING}[kind]
    lines = text.split("\n")
    if complete:
        assert lines[-1] == "", "file ends with a newline"
    else:   # the authors' NN log stops in the middle of a train() call: its last line has no newline yet
        assert rx.match(lines[-1])
    for ln in lines[:-1]:
        assert rx.match(ln.rstrip(" ") if kind != "nn" else ln), (kind, ln)
    return chart_parse(kind, text)


def test_the_grammar_is_the_references():
    imp = check("improvement", ref_text("improvement"))
    assert len(imp) == 65 and imp[0] == (0, 53, 145, 33, 58, 4)
    ben = check("benchmark", ref_text("benchmark"))
    assert len(ben) == 13 and ben[0][:3] == (0, 1, 0) and all(sum(r[1:2]) + r[3] + r[5] == 10 for r in ben)   # 10 games vs Random
    nn = check("nn", ref_text("nn"), complete=False)
    assert len(nn) == 66 and all(len(r) % 2 == 0 for r in nn) and nn[0][:2] == (3.49221, 0.632382)


def test_python_learn_loop_lines():
    L = importlib.import_module("alphazero-risk_amd.learn")
    gr = dict(count=250, draw=53, win=[145, 58], win_and_started=[33, 4])
    assert L.improvement_line(0, gr) == ref_first_line("improvement")   # the reference's first line
    r = dict(count=10, draw=1, win=[0, 9], win_and_started=[0, 4])
    s = dict(count=100, draw=0, win=[0, 100], win_and_started=[0, 50])
    assert L.benchmark_line(0, r, s) == ref_first_line("benchmark")
    first = ref_first_line("nn")
    vals = [float(x) for x in first.split(",")[:-1]]
    assert L.nn_training_line(list(zip(vals[0::2], vals[1::2]))) == first
    check("improvement", L.improvement_line(7, gr) + L.improvement_line(8, gr))
    check("benchmark", L.benchmark_line(3, r, s))
    check("nn", L.nn_training_line([(3.4922101497650146, 0.6323819756507874), (1e-7, 12345678.0)]) + L.nn_training_line([]))


def test_cpp_host_lines(tmp_path):
    host = os.path.join(ROOT, "alphazero-risk_amd", "host")
    csrc = os.path.join(ROOT, "alphazero-risk_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc])
    subprocess.check_call(["make", "-s", "-C", host])
    exe = str(tmp_path / "samples_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", host, os.path.join(ROOT, "tests", "helpers", "samples_probe.cpp"),
                           os.path.join(host, "azr_host.o"), "-o", exe, "-L", csrc, "-lazr_hip", "-Wl,-rpath," + csrc])
    out = subprocess.run([exe, "loglines", "0", "53", "145", "33", "58", "4", "0", "0", "0", "100", "50", "3.49221"],
                         capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0] + "\n" == ref_first_line("improvement")
    check("improvement", out[0] + "\n")
    check("benchmark", out[1] + "\n")
    check("nn", out[2] + "\n")
    assert out[2].startswith("3.49221, 1.16407, ")
