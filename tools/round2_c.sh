#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_cbam; rocprofv3 --kernel-trace --stats -d $O/prof_cbam -o cbam --output-format csv -- python3 $R/tools/cbam_bench.py > $O/prof_cbam.log 2>&1
find $O/prof_cbam -name "*kernel_trace.csv" -delete; tail -4 $O/prof_cbam.log
