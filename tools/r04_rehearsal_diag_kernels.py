#!/usr/bin/env python3
"""Cuts a rocprofv3 kernel trace of tools/r04_rehearsal_diag.py into its runs (marker: the scan kernel of torch.cumsum over 7777
floats, launched before each run) and prints, per run, how often each convolution-like kernel ran; then the kernels whose counts
differ between runs.   usage: r04_rehearsal_diag_kernels.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import re
import sys
from collections import Counter

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, f"no kernel trace under {d}"
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = ["R0_plain", "R0b_plain", "R1_modeB", "R0c_plain"]
segs, cur = [], None
for _, k in rows:
    if "scan" in k.lower():                     # a cumsum is one to three scan kernels: a new run starts at the first of them
        if cur is None or sum(cur.values()) > 50:
            cur = Counter()
            segs.append(cur)
        continue
    if cur is not None:
        cur[k] += 1
print(f"{len(rows)} kernel records, {len(segs)} runs found")


def short(k):
    k = re.sub(r"\(.*", "", k)
    return k[:110]


conv_like = re.compile(r"conv|igemm|gemm|Cijk|winograd|miopen|Im2|Col2|transpose|naive|gridwise|xdlops|batched", re.I)
per = []
for i, s in enumerate(segs):
    c = Counter()
    for k, n in s.items():
        if conv_like.search(k) and "lq::" not in k:
            c[short(k)] += n
    per.append(c)
    print(f"\n== {names[i] if i < len(names) else i}: {sum(s.values())} kernels, {sum(c.values())} convolution-like")
    for k, n in c.most_common(40):
        print(f"  {n:5d}  {k}")
print("\n== kernels whose count differs between the runs")
allk = set().union(*[set(c) for c in per]) if per else set()
for k in sorted(allk):
    counts = [c.get(k, 0) for c in per]
    if len(set(counts)) > 1:
        print(f"  {counts}  {k}")
