# A/B of library builds on isolated GEMM shapes + the bench step: bash tools/ab_gemm_lib.sh lib1.so lib2.so ...
for L in "$@"; do echo LIB $L; export M3T_LIB_PATH=$PWD/$L
M3T_GEMM_X6W=0 python tools/gemm_one.py 0 1 9600 1536 1024 50 2>&1 | tail -1
python tools/gemm_one.py 0 1 9600 1536 1024 50 2>&1 | tail -1
python tools/gemm_one.py 0 0 9600 1024 1536 50 2>&1 | tail -1
python tools/gemm_one.py 1 0 1536 1024 9600 50 2>&1 | tail -1
python tools/gemm_one.py 0 1 9600 512 1024 50 2>&1 | tail -1
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --aux "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"])"
done
