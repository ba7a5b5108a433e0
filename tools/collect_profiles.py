#!/usr/bin/env python3
"""Copy the summaries of tools/profile_bench.sh runs (gpurun_out/prof_TAG/) into profiles/ (tracked) and refresh
profiles/pmc_traffic.json, the per-configuration HBM-side traffic bench.py reports as roofline.traffic.
    python tools/collect_profiles.py r02 g512_s100_t2:g512_s100_t2_b20_bf16 [TAG:KEY ...]
traffic per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB as rocprofv3 reports them; the factor 2 is the
gfx950 correction for wide coalesced streaming reads, MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1]
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(tj)) if os.path.exists(tj) else {}
    for spec in sys.argv[2:]:
        tag, key = spec.split(":")
        src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
        for name in ("kernel_stats.csv", "pmc_summary.csv", "bench_under_rocprof.json"):
            if os.path.exists(os.path.join(src, name)):
                shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", f"{rnd}_{tag}_{name}"))
        rows = []   # kernel,counter,dispatches,mean,min,max — kernel names may hold commas: split from the right
        for ln in open(os.path.join(src, "pmc_summary.csv")).read().splitlines()[1:]:
            k, c, n, mean, lo, hi = ln.rsplit(",", 5)
            rows.append({"kernel": k.strip('"'), "counter": c, "dispatches": n, "mean": mean})
        # dominant kernel = the one with the largest total FETCH_SIZE
        fetch = {r["kernel"]: float(r["mean"]) * int(r["dispatches"]) for r in rows if r["counter"] == "FETCH_SIZE"}
        dom = max(fetch, key=fetch.get)
        val = {r["counter"]: float(r["mean"]) for r in rows if r["kernel"] == dom}
        stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}
        avg_ns = next((float(v["AverageNs"]) for k, v in stats.items() if dom.split("<")[0] in k), None)
        traffic[key] = {"kernel": dom, "bytes_per_launch": (2 * val["FETCH_SIZE"] + val["WRITE_SIZE"]) * 1024,
                        "fetch_size_kib": val["FETCH_SIZE"], "write_size_kib": val["WRITE_SIZE"],
                        "mfma_busy_cycles_per_launch": val.get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_wave_quadcycles_per_launch": val.get("SQ_WAVE_CYCLES"),
                        "kernel_avg_ns_under_rocprof": avg_ns, "source": f"profiles/{rnd}_{tag}_pmc_summary.csv"}
        print(key, traffic[key])
    json.dump(traffic, open(tj, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
