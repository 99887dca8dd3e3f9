#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace -d $O/lag -o lag --output-format csv -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --aux "" > $O/lag_bench.log 2>&1
python3 $R/tools/launch_lag.py $O/lag > $O/launch_lag.txt 2>&1
rm -rf $O/lag
