#!/usr/bin/env python3
"""gpurun_out/prof_r05pmc/ (tools/profile_r05_pmc.sh) -> profiles/r05_sensevoice_c4_pmc.json: per kernel of the compiled configs[3]-shard
plan, averaged per launch: HBM-side bytes (FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes for wide reads, WRITE_SIZE as
reported; separate passes), matrix-core busy cycles, VALU-active quad-cycles, wave quad-cycles, LDS instructions and bank-conflict cycles."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r05pmc")
KEEP = ("igemm_rs_kernel", "igemm_rs_ks4_kernel", "igemm_as_kernel", "attention_flash_kernel", "layer_norm_reg_kernel", "dwconv1d_tlc_kernel")


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(src, sub)
    for dirpath, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(dirpath, f))):
                    k = short(r["Kernel_Name"])
                    if any(k.startswith(p) for p in KEEP):
                        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(per.items()):
    row = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    row["launches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in row:
        row["read_MB_corrected"] = round(row["FETCH_SIZE"] * 1024 * 2 / 1e6, 2)
    if "WRITE_SIZE" in row:
        row["write_MB"] = round(row["WRITE_SIZE"] * 1024 / 1e6, 2)
    if row.get("SQ_WAVE_CYCLES"):
        row["valu_active_over_wave_cycles"] = round(row.get("SQ_ACTIVE_INST_VALU", 0) / row["SQ_WAVE_CYCLES"], 3)
    out[k] = row
json.dump({"note": "rocprofv3 --pmc over tools/sensevoice_graph.py --compiled-only --configs c4 --layers 10 --runs 3 (32 x 10 s utterances, eager); three "
                   "passes: {SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT}, "
                   "FETCH_SIZE, WRITE_SIZE; averages per launch; FETCH_SIZE / WRITE_SIZE in KiB as reported, read_MB_corrected = FETCH_SIZE x 2 "
                   "(gfx950 counts a 128-byte read request as 64) -- an UPPER bound where a kernel's requests really are 64 bytes (the quantising loaders "
                   "and igemm_as_kernel read a row as 64-byte pieces: their f32 rows appear twice); taken before the FSMN kernel's XCD-contiguous block order; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles",
           "kernels": out}, open(os.path.join(ROOT, "profiles", "r05_sensevoice_c4_pmc.json"), "w"), indent=1)
for k, row in out.items():
    print(k, {c: row[c] for c in ("launches", "read_MB_corrected", "write_MB", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT") if c in row})
