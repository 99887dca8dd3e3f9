for prof in 0 1 2; do
  echo "== M3T_SCAN_PROF=$prof"
  M3T_SCAN_PROF=$prof SCAN_FLAGS=$1 ONLY="4x512,fusion (2x512)" timeout 200 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids
done
