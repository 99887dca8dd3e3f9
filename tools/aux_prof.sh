#!/bin/bash
# rocprofv3 kernel stats of one aux leg: usage  bash tools/aux_prof.sh <leg> <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$2
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
M3T_SCAN_LOCK=0 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o p --output-format csv -- python3 $R/bench.py --aux-child $1 > $O/prof_$1.log 2>&1
f=$(find $O/prof_$1 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$1.csv; rm -rf $O/prof_$1
