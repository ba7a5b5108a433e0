#!/usr/bin/env python3
"""Generates tests/golden/ref_logs.json from the OUTPUT files of the reference's own training runs
(/root/reference/python/log/azr-{improvement,benchmark,nn-training}-log.txt: data the authors committed, written by
alphazero_trainer.cpp:139,163 through GameResults' operator<< (game.cpp:227-235) and by the LOG_NN_TRAINING branch of
alphazero_nn.cpp, read back by python/src/log_chart.py).  Used by tests/test_log_grammar.py as grammar fixtures.

Run in the build container only:   python tests/golden/make_logs_golden.py"""
import json
import os

SRC = "/root/reference/python/log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_logs.json")


def main():
    out = {}
    for key, name in (("improvement", "azr-improvement-log.txt"), ("benchmark", "azr-benchmark-log.txt"), ("nn", "azr-nn-training-log.txt")):
        out[key] = {"file": "python/log/" + name, "text": open(os.path.join(SRC, name)).read()}
    json.dump(out, open(OUT, "w"), indent=1)
    print({k: len(v["text"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
