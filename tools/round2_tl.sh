#!/bin/bash
# kernel trace of a short bench run -> per-stream timeline of the last step (tools/timeline.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-tl}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$T; rocprofv3 --kernel-trace -d $O/prof_$T -o $T --output-format rocpd -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_${T}.log 2>&1
tail -c 400 $O/prof_${T}.log
db=$(find $O/prof_$T -name "*.db" | head -1); echo "db=$db"
[ -n "$db" ] && python3 $R/tools/timeline.py $db ${2:-60e3} > $O/timeline_$T.txt 2>&1; python3 $R/tools/timeline.py $db 2e3 > $O/timeline_${T}_fine.txt 2>&1; tail -5 $O/timeline_$T.txt
rm -rf $O/prof_$T
