for cfg in "X=1" "M3T_WEIGHT_AMAX=0" "M3T_SCAN_PREP_AHEAD=0" "M3T_WEIGHT_AMAX=0 M3T_SCAN_PREP_AHEAD=0"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux "" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
