#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_bench.sh into one text file."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"== {os.path.relpath(f, d)}")
    out += [l.rstrip() for l in open(f)]
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    out.append(f"== {os.path.relpath(f, d)} (sum over dispatches; dispatches per counter in brackets)")
    for k in acc:
        if "render_kernel" not in k and "trace_kernel" not in k and "convert" not in k: continue
        for c, v in sorted(acc[k].items()):
            out.append(f"{k:60s} {c:32s} {v:.6g} [{n[(k,c)]}]")
print("\n".join(out))
