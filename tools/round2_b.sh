#!/bin/bash
# A/B: bf16x6 backward scan and multi-stream weight gradients
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02b}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_ddp.py -q -m gpu -x -k "gru or persist or bench or c3 or c2 or sink or shared" > $O/pytest_$T.log 2>&1; tail -4 $O/pytest_$T.log
M3T_SCAN_PROF=1 python tools/scan_bench.py 2>&1 | head -16 > $O/scan_$T.log; cat $O/scan_$T.log
M3T_SCAN_BWD_X6=0 python tools/scan_bench.py 2>&1 | head -9 > $O/scan_${T}_fp32bwd.log; cat $O/scan_${T}_fp32bwd.log
for cfg in "M3T_SCAN_BWD_X6=0 M3T_WGRAD_STREAMS=1" "M3T_SCAN_BWD_X6=1 M3T_WGRAD_STREAMS=1" "M3T_SCAN_BWD_X6=1 M3T_WGRAD_STREAMS=2" "M3T_SCAN_BWD_X6=1 M3T_WGRAD_STREAMS=3" "M3T_SCAN_BWD_X6=0 M3T_WGRAD_STREAMS=2"; do
  echo "== $cfg"; env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels'])"
done 2>&1 | tee $O/ab_$T.log
