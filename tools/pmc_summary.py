#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel and counter, dispatch count and mean/min/max value.
    python tools/pmc_summary.py DIR [DIR ...]      (DIR = a rocprofv3 -d output directory)"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_][\w:]*(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    acc = defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("kernel,counter,dispatches,mean,min,max")
    for (k, c), v in sorted(acc.items(), key=lambda kv: (kv[0][1], -sum(kv[1]))):
        print(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f}")


if __name__ == "__main__":
    main()
