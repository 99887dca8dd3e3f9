#!/bin/bash
# split-K sweep over the GEMM shapes of a C3 step (tools/gemm_bench.py with M3T_GEMM_SPLITS forced); prints us per shape and split
cd $GRAFT_REPO_ROOT
for sp in 0 1 2 3 4 6 8 10 12 16 20 24 32; do
  echo "SPLITS=$sp"
  M3T_GEMM_SPLITS=$sp python tools/gemm_bench.py 2>/dev/null | grep -E "dW|dX|fwd" | grep -vE "fc2|N  128|N  384" | awk '{printf "%s_%s_%s_%s_%s %s\n", $1,$2,$3,$8,$10,$13}' 
done
