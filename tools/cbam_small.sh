#!/bin/bash
# CBAM small-map iteration loop: parity tests, then rocprofv3 kernel stats on the three small ResNet-18 stage shapes.
# usage: bash tools/cbam_small.sh <tag> [notest]
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T
mkdir -p $O; cd $R
if [ "$2" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cbam" > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
python tools/cbam_bench.py > $O/bench.log 2>&1; cat $O/bench.log
cd /tmp && export TMPDIR=/tmp
for sh in "128 14" "256 7" "512 4"; do set -- $sh
  CBAM_SHAPES="$1,$2" rocprofv3 --kernel-trace --stats -d $O/prof -o cb --output-format csv -- python3 $R/tools/cbam_bench.py > $O/prof_$1.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$1_$2.csv; rm -rf $O/prof
  echo "== $1 x $2"; python3 - $O/kernel_stats_$1_$2.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    n=r['Name']; i=n.find('cbam_'); print("  %-28s %3s  %9.1f us"%(n[i:i+26] if i>=0 else n[:26], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
