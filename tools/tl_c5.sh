#!/bin/bash
# kernel timeline of one training step of an aux leg (bench.py --aux-child <leg>, default c5) by stream: usage  bash tools/tl_c5.sh <tag> [thr_ns] [leg]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-c5tl}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl; M3T_SCAN_LOCK=0 rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format rocpd -- python3 $R/bench.py --aux-child ${3:-c5} > $O/prof_tl.log 2>&1
db=$(find $O/prof_tl -name "*.db" | head -1)
python3 $R/tools/timeline.py $db ${2:-20e3} > $O/${T}_timeline.txt 2>&1
rm -rf $O/prof_tl
