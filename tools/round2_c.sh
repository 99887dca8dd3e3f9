#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl; rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format rocpd -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_tl.log 2>&1
db=$(find $O/prof_tl -name "*.db" | head -1); python3 $R/tools/timeline.py $db 60e3 > $O/r02b_timeline.txt 2>&1; rm -rf $O/prof_tl
cat $O/r02b_timeline.txt
