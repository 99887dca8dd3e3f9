#!/bin/bash
# Round-6 measurement set on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of the same command, the two HBM-traffic
# PMC passes, one SQ (MFMA utilisation) pass, kernel stats of the secondary configs, the untraced scan timeline, the ring GEMM's counters and
# A/B, the backward scan's stamp table.  Outputs under gpurun_out/; the summaries are copied to profiles/ (tracked).
# usage: bash tools/profile_round6.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r06}
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err; tail -c 400 $O/bench_$T.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$T; rocprofv3 --kernel-trace --stats -d $O/prof_$T -o $T --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux "" > $O/prof_${T}_bench.log 2>&1
find $O/prof_$T -name "*kernel_trace.csv" -delete
f=$(find $O/prof_$T -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_c3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_$c.log 2>&1; f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_summary.py $f $c > $O/pmc_$c.json; rm -rf $O/pmc_$c; done
python3 $R/tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/${T}_pmc_traffic.json > /dev/null
rm -rf $O/pmc_sq; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_sq.log 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_mfma.py $f $O/${T}_pmc_mfma.json; rm -rf $O/pmc_sq
for c in c1 c2 c2bf16 c5 cbam resnet3d; do
  rm -rf $O/prof_$c; M3T_SCAN_LOCK=0 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o ${T}_$c --output-format csv -- python3 $R/bench.py --aux-child $c > $O/prof_$c.log 2>&1
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
cd $R
M3T_BENCH_SCAN_TIMELINE=1 M3T_BENCH_EVENTS=1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --aux "" 2>&1 >/dev/null | grep "# tl" > $O/${T}_scan_timeline_untraced.txt
# the ring GEMM: counters of the NT kernel (variant 3) on the verdict's shape, then the A/B table
cd /tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  rm -rf $O/pmc_ring; RING_SHAPES=0x1x9600x1536x1024 rocprofv3 --pmc $set --kernel-trace -d $O/pmc_ring -o p --output-format csv -- python3 $R/tools/ring_bench.py 3 1 > $O/pmc_ring.log 2>&1
  f=$(find $O/pmc_ring -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "sgemm" not in k: continue
    acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); n[k[:70]].add(r["Dispatch_Id"])
for k, d in acc.items():
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(d.items()): print("   %-34s %16.0f  per dispatch %14.0f" % (c, v, v / max(1, len(n[k]))))
PY
  rm -rf $O/pmc_ring
done > $O/${T}_gemm_pmc_ring_nt_9600x1536x1024.txt 2>&1
cd $R
python tools/ring_bench.py 0,3 3 2>&1 | grep -v amdgpu.ids > $O/${T}_ring_ab.txt
bash tools/scan_prof.sh 32 > $O/${T}_scan_stamps.txt 2>&1
ls $O | grep $T
