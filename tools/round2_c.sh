#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_faults.py -q -m gpu -x 2>&1 | tail -3
REPS=6 STEPS=30 bash tools/ab.sh lb "M3T_SCAN_LIGHT_BATCHED=0" "M3T_SCAN_LIGHT_BATCHED=1"
