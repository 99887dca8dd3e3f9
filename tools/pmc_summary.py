#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: launches, mean and total counter value."""
import collections
import csv
import json
import re
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
with open(path) as fh:
    for r in csv.DictReader(fh):
        if r.get("Counter_Name") != counter:
            continue
        m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
        a = acc[k]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {k: {"launches": n, "mean": v / n, "total": v, "mean_ns": t / n} for k, (v, n, t) in acc.items()}
top = sorted(out.items(), key=lambda kv: -kv[1]["total"])
top = top[:14] + [kv for kv in top[14:] if kv[0].startswith("gru_")]       # every scan kernel, whatever its share
print(json.dumps({"counter": counter, "kernels": dict(top)}, indent=1))
