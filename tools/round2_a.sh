#!/bin/bash
# full GPU test-suite + bench (with aux) + kernel trace (timeline) + in-kernel scan stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02a}
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_$T.log 2>&1; tail -4 $O/pytest_$T.log
python bench.py --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err; tail -c 1200 $O/bench_$T.json
M3T_SCAN_PROF=1 python tools/scan_bench.py > $O/scan_$T.log 2>&1; tail -30 $O/scan_$T.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$T; rocprofv3 --kernel-trace --stats -d $O/prof_$T -o $T --output-format csv rocpd -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_${T}.log 2>&1
ls -la $O/prof_$T/*/ 2>/dev/null | head; find $O/prof_$T -name "*kernel_trace.csv" -delete
db=$(find $O/prof_$T -name "*.db" | head -1); [ -n "$db" ] && python3 $R/tools/timeline.py $db 100e3 > $O/timeline_$T.txt 2>&1; tail -60 $O/timeline_$T.txt
find $O/prof_$T -name "*.db" -delete
