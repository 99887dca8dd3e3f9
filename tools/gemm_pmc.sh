#!/bin/bash
# PMC passes over ONE GEMM shape (tools/gemm_one.py): where the tile kernel's cycles go.   usage: bash tools/gemm_pmc.sh tA tB M N K
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1)); rm -rf $O/pmc_g$i
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_g$i -o p --output-format csv -- python3 $R/tools/gemm_one.py "$@" 3 > $O/pmc_g$i.log 2>&1
  f=$(find $O/pmc_g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "sgemm" not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-34s %14.0f" % (c, v))
PY
  rm -rf $O/pmc_g$i
done
