#!/usr/bin/env python3
"""Instruction mix of the largest loop of a kernel in a hipcc -S listing (developer tool).
usage: isa_loop.py file.s <mangled-name-prefix>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pref) and ":" in l.split()[0]][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
loops.sort(key=lambda x: -(x[1] - x[0]))
print("kernel lines", len(body), "loops (start, end, len):", [(a, b, b - a) for a, b in loops[:6]])
a, b = loops[0]
c = collections.Counter()
for l in body[a:b]:
    t = l.strip()
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    c[t.split()[0]] += 1
groups = collections.Counter()
for k, v in c.items():
    g = ("v_pk" if k.startswith("v_pk") else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else
         "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global", "buffer", "flat")) else "other")
    groups[g] += v
print("loop instrs", sum(c.values()), dict(groups))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("  %-28s %d" % (k, v))
