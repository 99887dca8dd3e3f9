#!/usr/bin/env python3
"""Merge the two tools/pmc_summary.py outputs (FETCH_SIZE pass, WRITE_SIZE pass) into profiles/*_pmc_traffic.json:
HBM/fabric bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports half of wide
coalesced reads -- MI355X_MICROARCH.md, HBM section).    usage: pmc_merge.py fetch.json write.json out.json"""
import json
import sys

fetch = json.load(open(sys.argv[1]))["kernels"]
write = json.load(open(sys.argv[2]))["kernels"]
out = {"how": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate passes) --kernel-trace --output-format csv -- "
              "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline; tools/pmc_summary.py per pass; tools/pmc_merge.py; "
              "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads, "
              "MI355X_MICROARCH.md section HBM).  For the persistent scan kernels the fetch count includes the "
              "agent-scope (sc1) polling loads of the h_t / dgh_t exchange, which miss L2 by design.",
       "kernels": {}}
for k, f in fetch.items():
    w = write.get(k)
    if w is None:
        continue
    out["kernels"][k] = {"launches": f["launches"], "fetch_kb_raw": round(f["mean"], 1), "write_kb": round(w["mean"], 1),
                         "hbm_bytes_per_launch": int((2 * f["mean"] + w["mean"]) * 1024), "mean_ns": round(f["mean_ns"])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
