#!/usr/bin/env python3
"""Was the GPU waiting for the HOST?  Joins a rocprofv3 --kernel-trace --hip-trace CSV pair on the correlation id: for every kernel
of one steady-state C3 step, lag = kernel start - end of the launching API call on the host.  A kernel whose lag is ~10 us or less
started as soon as the host had issued it (the GPU was idle or just freed: host-bound at that point); lags of milliseconds mean
the host is far ahead.  usage: launch_lag.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ht = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
api = {}
for r in csv.DictReader(open(ht)):
    if "Launch" in r["Function"] or "launch" in r["Function"]:
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"])
rows = []
for r in csv.DictReader(open(kt)):
    a = api.get(r["Correlation_Id"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], a, r.get("Stream_Id", "?")))
rows.sort()
loss = [i for i, r in enumerate(rows) if "va_loss_grad" in r[2]]
seg = rows[loss[-3]:loss[-2]]
base = seg[0][0]
print("one step: %d kernels, span %.3f ms" % (len(seg), (seg[-1][1] - base) / 1e6))
prev_end = {}
small = 0
for s, e, n, a, st in seg:
    if a is None:
        continue
    lag = (s - a[1]) / 1e3
    gap = (s - prev_end.get(st, s)) / 1e3
    prev_end[st] = e
    if lag < 30.0:
        small += 1
    if lag < 30.0 or "gru_persist" in n or "gru_solo" in n:
        print("%8.3f ms st%-3s lag %9.1f us  gap-on-stream %7.1f us  dur %7.1f us  %s" % ((s - base) / 1e6, st, lag, gap, (e - s) / 1e3, n.replace("(anonymous namespace)::", "")[:60]))
print("%d of %d kernels started within 30 us of their launch call" % (small, len(seg)))
