#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_faults.py -q -m gpu -x -k "persist or gru or bench or arena or poll or dead or second or waves" 2>&1 | tail -3
for v in 0 1; do echo "== M3T_SCAN_L2=$v"; M3T_SCAN_L2=$v python tools/scan_bench.py 2>&1 | grep -E "^4x512|^scorers|^fusion|^2x256" ; done
REPS=3 bash tools/ab.sh l2 "M3T_SCAN_L2=0" "M3T_SCAN_L2=1"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_$c.log 2>&1; f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_summary.py $f $c > $O/pmc_$c.json; rm -rf $O/pmc_$c; done
python3 $R/tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/l2_traffic.json > /dev/null
python3 -c "
import json
d=json.load(open('$O/l2_traffic.json'))['kernels']
for k,v in d.items():
    if 'persist' in k: print(k, v['launches'], 'bytes', v['hbm_bytes_per_launch'])"
