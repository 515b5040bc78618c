#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked files under profiles/:

  <tag>_frontend_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `python bench.py --no-model` (kernel_stats table, verbatim)
  <tag>_bench_plain.json              the JSON line of an unprofiled default `python bench.py`
  <tag>_bench_under_rocprof.json      the JSON line of the profiled run (its roofline.kernel_ms must agree with AverageNs above)
  <tag>_frontend_pmc.json             FETCH_SIZE / WRITE_SIZE / SQ counters of fe_main_kernel, averaged per launch
  <tag>_valu_rate.json                tools/valu_rate: measured VALU issue cost per wave-instruction (plain and packed f32)
  <tag>_sensevoice_{c3,c4}_compiled_kernel_stats.csv, <tag>_microbench.json, <tag>_qlinear_variants.json
  frontend_roofline.json              the constants bench.py copies into its roofline block: HBM bytes per launch (PMC) and the VALU
                                      work of one launch priced at the measured issue rate

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB and were
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B read requests at 64 B, so wide coalesced streaming reads are
doubled; WRITE_SIZE is taken as reported (it matches the kernel's known output bytes)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def counters(path, kernel="fe_main_kernel"):
    vals = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_frontend_kernel_stats.csv"))
    for name in ("bench_under_rocprof.json", "bench_plain.json"):
        rec = last_json(os.path.join(src, name))
        if rec:
            json.dump(rec, open(os.path.join(dst, tag + "_" + name), "w"))
    for c in ("c3", "c4"):
        p = os.path.join(src, "sv", c + "_compiled_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_sensevoice_%s_compiled_kernel_stats.csv" % (tag, c)))
    p = os.path.join(src, "yolo", "n64_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_yolo_shaped_n64_kernel_stats.csv" % tag))
    for a, b in (("yolo_n64.json", "_yolo_shaped_n64.json"), ("yolo_table.json", "_yolo_shaped_n64_table.json"), ("yolo_lifted_n64.json", "_yolo26seg_lifted_n64.json"),
                 ("yolo_lifted_table.json", "_yolo26seg_lifted_n64_table.json"), ("conv_integer_i8.json", "_conv_integer_i8.json"),
                 ("conv_integer_f32.json", "_conv_integer_f32.json"), ("recip_check.json", "_recip_check.json"), ("microbench.json", "_microbench.json"), ("qlinear.json", "_qlinear_variants.json"), ("valu_rate.json", "_valu_rate.json"),
                 ("attention_bench.json", "_attention_bench.json"), ("attention_stamps.txt", "_attention_stamps.txt"),
                 ("rs_stamps.txt", "_rs_stamps.txt"), ("rs_bench.txt", "_rs_bench.txt"), ("l2bw.txt", "_l2bw.txt")):
        if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)) > 2:
            shutil.copy(os.path.join(src, a), os.path.join(dst, tag + b))
    # attention kernels: matrix-core busy cycles and VALU-active cycles from ONE pass, per kernel, averaged per launch
    apath = os.path.join(src, "pmc_attn", "attn_counter_collection.csv")
    if os.path.exists(apath):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(apath) as f:
            for r in csv.DictReader(f):
                m = re.search(r"attention\w*_kernel(<[^>]*>)?", r["Kernel_Name"])
                if m:
                    per[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rep = {}
        for k, cs in per.items():
            row = {c: sum(v) / len(v) for c, v in cs.items()}
            row["launches"] = len(next(iter(cs.values())))
            if row.get("SQ_BUSY_CYCLES"):
                # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs of every CU; SQ_BUSY_CYCLES per SE: report raw + the ratio
                row["mfma_busy_over_valu_active"] = round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, row.get("SQ_ACTIVE_INST_VALU", 1.0)), 3)
            rep[k] = row
        json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU in one pass over "
                           "tools/attention_bench.py --only default (32 x 171, 1 x 504, 8 x 171, 64 x 171 rows); averages per launch; SQ_WAVE_CYCLES / "
                           "SQ_ACTIVE_INST_* are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles (MI355X_MICROARCH.md)", "kernels": rep},
                  open(os.path.join(dst, tag + "_attention_pmc.json"), "w"), indent=1)
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        got = counters(os.path.join(src, "pmc_" + c, "bench_counter_collection.csv"))
        if got:
            pmc[c + "_KiB_avg"] = got[0].get(c)
            pmc["launches_" + c] = got[1].get(c)
    sq = counters(os.path.join(src, "pmc_sq", "bench_counter_collection.csv"))
    if sq:
        pmc.update({k: v for k, v in sq[0].items()})
    if "FETCH_SIZE_KiB_avg" in pmc and "WRITE_SIZE_KiB_avg" in pmc:
        pmc["read_bytes_corrected"] = int(pmc["FETCH_SIZE_KiB_avg"] * 1024 * 2)  # gfx950: FETCH_SIZE is half of wide reads
        pmc["write_bytes"] = int(pmc["WRITE_SIZE_KiB_avg"] * 1024)
        pmc["hbm_bytes_per_launch"] = pmc["read_bytes_corrected"] + pmc["write_bytes"]
    bench = last_json(os.path.join(src, "bench_plain.json")) or last_json(os.path.join(src, "bench_under_rocprof.json"))
    batch = bench["config"]["batch_per_gpu"]
    json.dump({"tag": tag, "batch": batch, "kernel": "fe_main_kernel", "counters": pmc,
               "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}, open(os.path.join(dst, tag + "_frontend_pmc.json"), "w"), indent=1)
    # VALU work of one launch at the measured issue cost.  The ISA of the pass loop (tools/isa_loop.py) holds 416 packed-f32 and 463
    # plain VALU instructions (round 4's last form; 519 and 448 before it); SQ_INSTS_VALU counts both as one.  Cost per wave-instruction and SIMD from tools/valu_rate (4 waves
    # per SIMD, the saturated regime): plain = the mean of v_add / v_mul / the fma+add pair, packed = the mean of the three v_pk ops.
    roof = {"batch": batch, "kernel": "fe_main_kernel", "source": "profiles/%s_frontend_pmc.json, profiles/%s_valu_rate.json" % (tag, tag)}
    if "hbm_bytes_per_launch" in pmc:
        roof["hbm_bytes_per_launch"] = pmc["hbm_bytes_per_launch"]
    vr = last_json_file(os.path.join(src, "valu_rate.json"))
    if vr and "SQ_INSTS_VALU" in pmc:
        plain = sum(vr[k]["wps4"]["ns_per_instr_per_simd"] for k in ("v_add_f32", "v_mul_f32", "v_fma_f32+v_add_f32_pair")) / 3
        packed = sum(vr[k]["wps4"]["ns_per_instr_per_simd"] for k in ("v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32")) / 3
        n = pmc["SQ_INSTS_VALU"]
        n_packed, n_plain = n * 416 / 879.0, n * 463 / 879.0
        simds = 1024
        floor_ms = (n_plain * plain + n_packed * packed) / simds * 1e-6
        lane_ops = n_plain * 64 + n_packed * 128
        roof.update({"valu_wave_instructions_per_launch": n, "valu_packed_fraction_isa": round(416 / 879.0, 4),
                     "valu_ns_per_plain_instr_per_simd": round(plain, 4), "valu_ns_per_packed_instr_per_simd": round(packed, 4),
                     "valu_issue_floor_ms": round(floor_ms, 5), "valu_lane_ops_per_launch": int(lane_ops),
                     # the rate at which the chip issues this kernel's own mix of plain and packed f32 lane-operations
                     "valu_peak_lane_ops_per_s": lane_ops / (floor_ms * 1e-3),
                     "valu_peak_source": "tools/valu_rate on this box: %.2f ns per plain and %.2f ns per packed-f32 wave-instruction and SIMD, "
                                         "applied to the kernel's instruction mix" % (plain, packed)})
    json.dump(roof, open(os.path.join(dst, "frontend_roofline.json"), "w"), indent=1)
    print(json.dumps(roof, indent=1))


def last_json_file(path):
    try:
        return json.load(open(path))
    except Exception:
        return None


if __name__ == "__main__":
    main()
