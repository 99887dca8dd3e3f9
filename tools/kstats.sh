#!/bin/bash
# rocprofv3 kernel stats of `bench.py --steps 10` under the caller's environment; prints the top kernels.  usage: tools/kstats.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-ks}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks_$T; rocprofv3 --kernel-trace --stats -d $O/ks_$T -o $T --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --aux "" > $O/ks_$T.log 2>&1
find $O/ks_$T -name "*kernel_trace.csv" -delete
f=$(find $O/ks_$T -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step %.3f' % (tot / 1e6 / 13))
for r in rows[:28]:
    print('%-100s calls/step %6.1f  ms/step %7.3f  avg %8.1f us' % (r['Name'][:100], int(r['Calls']) / 13, float(r['TotalDurationNs']) / 1e6 / 13, float(r['AverageNs']) / 1e3))
PY
