#!/usr/bin/env python3
"""Per-pass timeline of an arena run from a rocprofv3 kernel trace (csv): kernel durations, grids and the gaps between them.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/arena_trace -- python3 tools/arena_bench.py
    python3 tools/arena_trace_summary.py gpurun_out/arena_trace/*/*_kernel_trace.csv"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_arena_step")]
    print(f"{len(rows)} kernels, {len(steps)} arena steps")
    per = collections.defaultdict(list)
    for r in rows[steps[0]:]:
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 1
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:24]
        per[(name, r["Queue_Id"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // max(wg, 1)))
    for k, v in sorted(per.items()):
        d = [x[0] for x in v]
        g = [x[1] for x in v]
        print(f"{k[0]:26s} queue {k[1]:>3s}: {len(v):6d} launches, mean {sum(d) / len(d) / 1e3:8.1f} us, total {sum(d) / 1e9:6.2f} s, workgroups mean {sum(g) / len(g):6.1f} max {max(g)}")
    period = [int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in zip(steps, steps[1:])]
    period.sort()
    print(f"pass period: median {period[len(period) // 2] / 1e3:.1f} us, mean {sum(period) / len(period) / 1e3:.1f} us")
    # time per pass in which no kernel runs at all
    busy_end, idle = 0, 0
    for r in rows[steps[0]:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if busy_end and s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end, e)
    print(f"no kernel running: {idle / 1e9:.2f} s = {idle / len(steps) / 1e3:.1f} us per pass")
    a = steps[len(steps) // 2]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:a + 24]:
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 1
        print(f"  {r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:24]:26s} q{r['Queue_Id']} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us  {int(r['Grid_Size_X']) // max(wg, 1)} wgs")


if __name__ == "__main__":
    main()
