#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.

    python tools/kres.py quant.hip igemm_as        # kernels of lele_amd/csrc/quant.hip whose (demangled) name contains `igemm_as`
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lele_amd import build as B  # noqa: E402

src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = [B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +([A-Za-z ]+): (.+?) \[-Rpass", line) or re.search(r"remark: +([A-Za-z ]+): (.+?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+: +([A-Za-z ]+?): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if pat in name:
        print("%-70s vgpr %4s agpr %3s spill %3s lds %6s occ %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill", r.get("ScratchSize [bytes/lane]")),
                                                                   r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
