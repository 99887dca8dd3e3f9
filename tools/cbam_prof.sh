#!/bin/bash
# rocprofv3 kernel stats of the CBAM gate on one stage shape: usage  bash tools/cbam_prof.sh <C> <HW> <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$3
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
CBAM_SHAPES="$1,$2" rocprofv3 --kernel-trace --stats -d $O/prof -o cb --output-format csv -- python3 $R/tools/cbam_bench.py > $O/bench.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$1_$2.csv; rm -rf $O/prof
