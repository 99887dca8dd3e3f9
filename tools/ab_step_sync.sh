python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_step or c5_affwild or no_grad_forward" 2>&1 | tail -5
for i in 1 2 3; do
  for v in 0 1; do
    echo "SYNC=$v: $(M3T_STEP_SYNC=$v M3T_AUX_STEPS=20 python bench.py --aux-child c5 2>/dev/null | grep -o '"aux": "c5[a-z_]*"[^}]*"ms_per_step": [0-9.]*' | sed 's/"workload.*"ms_per_step"/ms/')"
  done
done
