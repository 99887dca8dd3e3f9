#!/bin/bash
# A/B harness: bench.py under several env configurations, interleaved, REPS times each; prints min / median ms per step.
# usage: tools/ab.sh tag "ENV1=a ENV2=b" "ENV1=c" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
REPS=${REPS:-3}; STEPS=${STEPS:-30}
cd $R
: > $O/ab_$T.log
for r in $(seq $REPS); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --aux "" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('cfg$i', d['ms_per_step'], k.get('gru_persist_fwd_kernel',{}).get('ms_per_step'), k.get('gru_persist_bwd_kernel',{}).get('ms_per_step'))" >> $O/ab_$T.log
  done
done
python - "$@" <<PY
import sys, collections, statistics
rows=collections.defaultdict(list)
for l in open("$O/ab_$T.log"):
    p=l.split(); rows[p[0]].append([float(x) if x!='None' else 0 for x in p[1:]])
for i,cfg in enumerate(sys.argv[1:],1):
    v=rows['cfg%d'%i]
    print("%-60s step min %.3f med %.3f | fwd scans %.2f bwd scans %.2f"%(cfg, min(x[0] for x in v), statistics.median(x[0] for x in v), statistics.median(x[1] for x in v), statistics.median(x[2] for x in v)))
PY
