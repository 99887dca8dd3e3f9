#!/bin/bash
# the two HBM-traffic PMC passes of the bench command only -> gpurun_out/<tag>_pmc_traffic.json      usage: bash tools/pmc_traffic.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_$c.log 2>&1; f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_summary.py $f $c > $O/pmc_$c.json; rm -rf $O/pmc_$c; done
python3 $R/tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/${T}_pmc_traffic.json > /dev/null
