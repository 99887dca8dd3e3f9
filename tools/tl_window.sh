#!/bin/bash
# full kernel timeline (every kernel >= 2 us) of a window of one C3 step: usage  bash tools/tl_window.sh <tag> <lo_ms> <hi_ms>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl; rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format rocpd -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_tl.log 2>&1
db=$(find $O/prof_tl -name "*.db" | head -1); python3 $R/tools/timeline.py $db 2e3 $2 $3 > $O/timeline_$2_$3.txt 2>&1; python3 $R/tools/timeline.py $db 2e3 9.5 16.5 > $O/timeline_fwd.txt 2>&1; rm -rf $O/prof_tl
