#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/dag_bench.py: do kernels overlap in time?  The run replays the same plan first as a
linear graph, then as a DAG; for each contiguous burst of kernel activity (a graph replay) report its length, the sum of its kernels'
durations and the time during which two or more kernels were in flight."""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print(json.dumps({"error": "no kernel_trace.csv under %s" % root}))
        return
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    bursts, cur = [], []
    for k in rows:
        if cur and k[0] - max(e for _s, e, _n in cur) > 200_000:   # a gap of 0.2 ms: another replay
            bursts.append(cur)
            cur = []
        cur.append(k)
    if cur:
        bursts.append(cur)
    out = []
    for b in bursts:
        if len(b) < 100:
            continue
        ev = sorted([(s, 1) for s, _e, _n in b] + [(e, -1) for _s, e, _n in b])
        depth, last, over, busy = 0, ev[0][0], 0, 0
        for t, d in ev:
            if depth >= 2:
                over += t - last
            if depth >= 1:
                busy += t - last
            depth += d
            last = t
        out.append({"kernels": len(b), "span_ms": round((max(e for _s, e, _n in b) - b[0][0]) / 1e6, 4), "sum_of_kernel_ms": round(sum(e - s for s, e, _n in b) / 1e6, 4),
                    "busy_ms": round(busy / 1e6, 4), "two_or_more_in_flight_ms": round(over / 1e6, 4)})
    # the last replays are the timed ones: linear graphs come first in dag_bench, DAG graphs after
    print(json.dumps({"trace": os.path.basename(files[0]), "bursts_of_100_plus_kernels": out[-16:]}, indent=0))


if __name__ == "__main__":
    main()
