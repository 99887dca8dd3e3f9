#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_faults.py -q -m gpu -x -k "arena or persist or gru or bench or fault or dead or second" 2>&1 | tail -4
REPS=3 bash tools/ab.sh arena "M3T_SCAN_ARENA=0" "M3T_SCAN_ARENA=1"
