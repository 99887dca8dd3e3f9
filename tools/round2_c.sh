#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_bf16.py -q -m gpu -x -k "conv or tcn or c2 or c1 or vggm" 2>&1 | tail -5
python bench.py --aux-child c1,c2,c2bf16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['aux'], d['ms_per_step'], d['clips_per_s'], d['alg_tflops'])"
