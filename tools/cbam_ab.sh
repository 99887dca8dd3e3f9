#!/bin/bash
# kernel times of the CBAM gate under several library builds (M3T_LIB_PATH): usage  bash tools/cbam_ab.sh <tag> lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for sh in "64 28" "128 14" "256 7" "512 4"; do set -- $sh
    export M3T_LIB_PATH=$R/m3f.pytorch_amd/lib/$lib CBAM_SHAPES="$1,$2"
    rocprofv3 --kernel-trace --stats -d $O/prof -o cb --output-format csv -- python3 $R/tools/cbam_bench.py > $O/log 2>&1
    f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
    python3 - $f "$lib $1x$2" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    n=r['Name']; i=n.find('cbam_')
    if i>=0: out.append("%s %.1f"%(n[i+5:i+9].strip('_k<('), float(r['AverageNs'])/1e3))
print(sys.argv[2], ' | '.join(out[:7]))
PY
    rm -rf $O/prof
  done
done
