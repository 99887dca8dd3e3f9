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
# algorithmic bytes of the scan launches of the C3 step (32 clips x 300 frames; DESIGN.md section 5: backward 48 B, forward 36 B per
# (row, unit, step) and scan) -> traffic / algorithmic per launch type
B_, T_ = 32, 300
ALG = {"gru_persist_fwd6_kernel<4, true, 2>": 36 * B_ * 512 * 4 * T_, "gru_persist_fwd6_kernel<4, true, 1>": 36 * B_ * 512 * 2 * T_,
       "gru_persist_fwd6_kernel<2, true, 2>": 36 * B_ * 256 * 2 * T_, "gru_persist_fwd6_kernel<2, true, 1>": 36 * B_ * 256 * 2 * T_,
       "gru_persist_bwd3q_kernel<2, false>": 48 * B_ * 256 * 2 * T_, "gru_persist_bwd3p_kernel<4>": 48 * B_ * 512 * 2 * T_,
       "gru_persist_bwd6_kernel<2>": 48 * B_ * 256 * 2 * T_, "gru_solo_fwd_kernel": 36 * B_ * 128 * 4 * T_, "gru_solo_bwd_kernel": 48 * B_ * 128 * 4 * T_,
       # bwd3q<4>: two fusion-level launches (2 scans) + two gru_v | gru_a launches (4 scans) per step: the mean launch
       "gru_persist_bwd3q_kernel<4, false>": 48 * B_ * 512 * 3 * T_}
for k, v in out["kernels"].items():
    if k in ALG:
        v["algorithmic_bytes_per_launch"] = ALG[k]
        v["traffic_over_algorithmic"] = round(v["hbm_bytes_per_launch"] / ALG[k], 3)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
