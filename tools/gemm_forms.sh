# the four operand forms on the shapes of the C3 step: bash tools/gemm_forms.sh
for X in 0 1; do echo "M3T_GEMM_X6W=$X"; export M3T_GEMM_X6W=$X
python tools/gemm_one.py 0 1 9600 1536 1024 50 2>&1 | tail -1
python tools/gemm_one.py 0 0 9600 1024 1536 50 2>&1 | tail -1
python tools/gemm_one.py 1 0 1536 1024 9600 50 2>&1 | tail -1
python tools/gemm_one.py 1 0 1536 512 9600 50 2>&1 | tail -1
python tools/gemm_one.py 0 1 9600 512 1024 50 2>&1 | tail -1
done
