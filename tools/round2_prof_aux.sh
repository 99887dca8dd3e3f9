#!/bin/bash
# rocprofv3 kernel stats of the secondary configs (bench.py --aux-child): profiles/r02_{c1,c2,c5}_kernel_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in ${1:-c1 c2 c2bf16 c5}; do
  rm -rf $O/prof_$c; rocprofv3 --kernel-trace --stats -d $O/prof_$c -o $c --output-format csv -- python3 $R/bench.py --aux-child $c > $O/prof_$c.log 2>&1
  grep '"aux"' $O/prof_$c.log | cut -c1-200
  find $O/prof_$c -name "*kernel_trace.csv" -delete
done
