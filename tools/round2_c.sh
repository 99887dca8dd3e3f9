#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in 0 1 2; do echo "== BWD6_VAR=$v"; M3T_SCAN_BWD6_VAR=$v M3T_SCAN_PROF=1 python tools/scan_bench.py 2>&1 | grep -A2 -E "^fusion|^4x512|^2x256" | grep -v "^--" ; done 2>&1 | tee $O/var_c.log
