#!/bin/bash
# End-of-round check on the GPU box (through gpurun): the full GPU test suite, the default bench line, and rocprofv3 kernel stats of the
# C5 and CBAM secondary legs (the ones that moved last).  Outputs under gpurun_out/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2000 python -m pytest tests -q -m gpu -x > $O/pytest_final.log 2>&1; tail -2 $O/pytest_final.log
python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 200 $O/bench_final.json; echo
cd /tmp && export TMPDIR=/tmp
for c in c5 cbam; do rm -rf $O/prof_$c; rocprofv3 --kernel-trace --stats -d $O/prof_$c -o r04_$c --output-format csv -- python3 $R/bench.py --aux-child $c > $O/prof_$c.log 2>&1; find $O/prof_$c -name "*kernel_trace.csv" -delete; done
ls $O/prof_c5/*/ 2>/dev/null | head -3
