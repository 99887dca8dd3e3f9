#!/bin/bash
# one step timeline of bench.py under rocprofv3 -> gpurun_out/<tag>_timeline.txt     usage: bash tools/tl.sh <tag> [thr_ns] [lo_ms hi_ms]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-tl}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl; rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format rocpd -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_tl.log 2>&1
db=$(find $O/prof_tl -name "*.db" | head -1)
if [ -n "$db" ]; then
  python3 $R/tools/timeline.py $db ${2:-60e3} > $O/${T}_timeline.txt 2>&1
  [ -n "$4" ] && python3 $R/tools/timeline.py $db 0 $3 $4 > $O/${T}_window.txt 2>&1
  [ -n "$6" ] && python3 $R/tools/timeline.py $db 0 $5 $6 > $O/${T}_window2.txt 2>&1
fi
rm -rf $O/prof_tl
