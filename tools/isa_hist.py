#!/usr/bin/env python3
"""Instruction histogram of the kernels in a hipcc -S listing (developer tool).
usage: isa_hist.py file.s [name-substring] [top]"""
import collections
import re
import sys

def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    cur = None
    hist = {}
    for ln in lines:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            cur = m.group(1)
            hist[cur] = collections.Counter()
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        t = ln.strip()
        if not t or t[0] in ".;/" or t.endswith(":"):
            continue
        hist[cur][t.split()[0]] += 1
    for name, c in hist.items():
        if pat not in name or not c:
            continue
        print(name, "total", sum(c.values()))
        groups = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_pk"): groups["v_pk"] += v
            elif k.startswith("v_mfma"): groups["mfma"] += v
            elif k.startswith("v_"): groups["valu"] += v
            elif k.startswith("ds_"): groups["lds"] += v
            elif k.startswith(("global_", "buffer_", "flat_", "scratch_")): groups["vmem"] += v
            elif k.startswith("s_"): groups["salu"] += v
            else: groups["other"] += v
        print("   groups:", dict(groups))
        for k, v in c.most_common(top):
            print("    %-28s %d" % (k, v))

if __name__ == "__main__":
    main()
