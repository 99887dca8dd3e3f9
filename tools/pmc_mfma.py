#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (counter_collection.csv) with
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16
SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE.
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed
  over the 8 XCDs -- MI355X_MICROARCH.md, DVFS give-back); busy cycles are summed over all SIMDs (= 32 x N for a 32x32x16 bf16 MFMA,
  16 x N for 16x16x32).  usage: pmc_mfma.py counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
            acc[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, c in acc.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[k] = {"launches": cnt[k], "mean_us": round(c["_ns"] / max(1, cnt[k]) / 1e3, 1),
              "mfma_busy_cycles": busy, "kernel_cycles": cyc,
              "mfma_util": round(busy / (1024.0 * cyc), 4) if cyc else None,
              "clock_GHz": round(cyc / c["_ns"], 3) if c["_ns"] else None,
              "mops_bf16": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"), "mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
              "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_any": c.get("SQ_WAIT_ANY"), "active_inst_any": c.get("SQ_ACTIVE_INST_ANY")}
top = dict(sorted(out.items(), key=lambda kv: -(kv[1]["mean_us"] * kv[1]["launches"]))[:16])
json.dump({"how": __doc__, "kernels": top}, open(sys.argv[2], "w"), indent=1)
for k, v in top.items():
    print("%-52s n=%4d %9.1f us  mfma_util %s  clock %s GHz" % (k[:52], v["launches"], v["mean_us"], v["mfma_util"], v["clock_GHz"]))
