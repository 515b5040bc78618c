#!/usr/bin/env python3
"""Per-dispatch view of a recorded forward from a rocprofv3 kernel trace (tools/ktrace.sh): takes the LAST `--runs` replays of a graph of
`--nodes` kernels (the tail of the trace), averages every position of the graph over the replays and prints duration, gap to the previous
dispatch's end, and -- with a second trace at half the batch -- the fixed part 2 t(b/2) - t(b) of every dispatch.

    python tools/ktrace_graph.py gpurun_out/ktrace_b64/k_kernel_trace.csv --nodes 192 --runs 10 [--half gpurun_out/ktrace_b32/k_kernel_trace.csv]"""
import argparse
import csv
import json
import re


def load(path, nodes, runs):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-nodes * runs:]
    pos = []
    for i in range(nodes):
        rs = [tail[k * nodes + i] for k in range(runs)]
        names = {r["Kernel_Name"] for r in rs}
        assert len(names) == 1, (i, names)
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / runs / 1e3
        gap = 0.0
        if i:
            gap = sum(int(tail[k * nodes + i]["Start_Timestamp"]) - max(int(tail[k * nodes + j]["End_Timestamp"]) for j in range(max(0, i - 4), i)) for k in range(runs)) / runs / 1e3
        g = rs[0]
        pos.append({"i": i, "name": re.sub(r"^void |lele::|\(anonymous namespace\)::|\(.*", "", g["Kernel_Name"])[:60], "us": dur, "gap_us": gap,
                    "grid": int(g.get("Grid_Size_X", g.get("Grid_Size", 0)) or 0), "wg": int(g.get("Workgroup_Size_X", g.get("Workgroup_Size", 0)) or 0)})
    span = sum(int(tail[(k + 1) * nodes - 1]["End_Timestamp"]) - int(tail[k * nodes]["Start_Timestamp"]) for k in range(runs)) / runs / 1e3
    return pos, span


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--nodes", type=int, required=True)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--half", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    pos, span = load(a.trace, a.nodes, a.runs)
    busy = sum(p["us"] for p in pos)
    rec = {"span_us": round(span, 1), "sum_of_kernel_us": round(busy, 1), "sum_of_positive_gaps_us": round(sum(max(0.0, p["gap_us"]) for p in pos), 1), "nodes": a.nodes}
    if a.half:
        hp, hspan = load(a.half, a.nodes, a.runs)
        for p, h in zip(pos, hp):
            p["half_us"] = h["us"]
            p["fixed_us"] = 2 * h["us"] - p["us"]
        rec.update({"half_span_us": round(hspan, 1), "half_sum_of_kernel_us": round(sum(h["us"] for h in hp), 1), "sum_fixed_us": round(sum(p["fixed_us"] for p in pos), 1)})
    print(json.dumps(rec))
    key = (lambda p: -p["fixed_us"]) if a.half else (lambda p: -p["us"])
    for p in sorted(pos, key=key)[:a.top]:
        print("%4d %-60s grid %8d  %8.1f us  gap %6.1f" % (p["i"], p["name"], p["grid"], p["us"], p["gap_us"]) + ("  half %8.1f fixed %7.1f" % (p["half_us"], p["fixed_us"]) if a.half else ""))
    if a.out:
        rec["dispatches"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items()} for p in pos]
        json.dump(rec, open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()
