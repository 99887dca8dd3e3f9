#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun): GPU tests, bench line, rocprofv3 kernel stats of the same
# command, and the two PMC passes behind roofline.traffic.  Outputs under gpurun_out/; copy the summaries to profiles/.
# usage: bash tools/profile_round.sh <tag> [nopmc]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${1:-r01}
cd $R && timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_$T.log 2>&1; tail -2 $O/pytest_$T.log
python bench.py --steps 20 --warmup 3 > $O/bench_$T.json 2> $O/bench_$T.err; tail -c 300 $O/bench_$T.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$T; rocprofv3 --kernel-trace --stats -d $O/prof_$T -o $T --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_${T}_bench.log 2>&1
find $O/prof_$T -name "*kernel_trace.csv" -delete
if [ "$2" != "nopmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1; f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_summary.py $f $c > $O/pmc_$c.json; rm -rf $O/pmc_$c; done
fi
