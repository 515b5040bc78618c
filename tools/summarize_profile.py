#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_frontend.sh on the GPU box) into the tracked files
under profiles/:  <tag>_frontend_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py`),
<tag>_frontend_pmc.json (FETCH_SIZE / WRITE_SIZE per kernel, raw and corrected) and frontend_hbm_traffic.json
(the per-launch HBM bytes bench.py reports as roofline.traffic).

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB and were
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B read requests at 64 B, so wide coalesced
streaming reads are doubled; WRITE_SIZE is taken as reported (it matches the kernel's known output bytes)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_frontend_kernel_stats.csv"))
    for name in ("bench_under_rocprof.json", "bench_plain.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            line = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if line:
                open(os.path.join(dst, tag + "_" + name), "w").write(line[-1] + "\n")
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        with open(os.path.join(src, "pmc_" + c, "bench_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            short = "fe_main_kernel" if "fe_main_kernel" in k else "fe_frame_sum_kernel" if "fe_frame_sum" in k else None
            if short:
                pmc.setdefault(short, {})[c + "_KiB_avg"] = sum(v) / len(v)
                pmc[short]["launches_" + c] = len(v)
    for k, d in pmc.items():
        if "FETCH_SIZE_KiB_avg" not in d or "WRITE_SIZE_KiB_avg" not in d:
            continue
        d["read_bytes_corrected"] = int(d["FETCH_SIZE_KiB_avg"] * 1024 * 2)  # gfx950: FETCH_SIZE is half of wide reads
        d["write_bytes"] = int(d["WRITE_SIZE_KiB_avg"] * 1024)
        d["hbm_bytes_per_launch"] = d["read_bytes_corrected"] + d["write_bytes"]
    bench = json.loads(open(os.path.join(dst, tag + "_bench_plain.json")).read())
    batch = bench["config"]["batch_per_gpu"]
    out = {"tag": tag, "batch": batch, "kernels": pmc,
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
    json.dump(out, open(os.path.join(dst, tag + "_frontend_pmc.json"), "w"), indent=1)
    json.dump({"batch": batch, "kernel": "fe_main_kernel", "bytes_per_launch": pmc["fe_main_kernel"]["hbm_bytes_per_launch"],
               "source": "profiles/%s_frontend_pmc.json" % tag},
              open(os.path.join(dst, "frontend_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
