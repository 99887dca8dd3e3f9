#!/bin/bash
# Round-5 measurement set on the GPU box (run through gpurun): GPU tests, bench line, rocprofv3 kernel stats of the same command,
# the two HBM-traffic PMC passes, one SQ (MFMA utilisation) pass, kernel stats of the secondary configs, a step timeline.
# Outputs under gpurun_out/; the summaries are copied to profiles/ (tracked).   usage: bash tools/profile_round5.sh <tag> [notest]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r05}
cd $R
if [ "$2" != "notest" ]; then timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_$T.log 2>&1; tail -3 $O/pytest_$T.log; fi
python bench.py --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err; tail -c 600 $O/bench_$T.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$T; rocprofv3 --kernel-trace --stats -d $O/prof_$T -o $T --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux "" > $O/prof_${T}_bench.log 2>&1
find $O/prof_$T -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_$c.log 2>&1; f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_summary.py $f $c > $O/pmc_$c.json; rm -rf $O/pmc_$c; done
python3 $R/tools/pmc_merge.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/${T}_pmc_traffic.json > /dev/null
rm -rf $O/pmc_sq; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --aux "" > $O/pmc_sq.log 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); python3 $R/tools/pmc_mfma.py $f $O/${T}_pmc_mfma.json; rm -rf $O/pmc_sq
for c in c1 c2 c2bf16 c5 cbam; do
  rm -rf $O/prof_$c; rocprofv3 --kernel-trace --stats -d $O/prof_$c -o ${T}_$c --output-format csv -- python3 $R/bench.py --aux-child $c > $O/prof_$c.log 2>&1
  find $O/prof_$c -name "*kernel_trace.csv" -delete
done
rm -rf $O/prof_tl; rocprofv3 --kernel-trace -d $O/prof_tl -o tl --output-format rocpd -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --aux "" > $O/prof_tl.log 2>&1
db=$(find $O/prof_tl -name "*.db" | head -1); [ -n "$db" ] && python3 $R/tools/timeline.py $db 60e3 > $O/${T}_timeline.txt 2>&1; rm -rf $O/prof_tl
for sh in "64 28" "128 14" "256 7" "512 4"; do set -- $sh; CBAM_SHAPES="$1,$2" rocprofv3 --kernel-trace --stats -d $O/prof_cb -o cb --output-format csv -- python3 $R/tools/cbam_bench.py > $O/prof_cb_$1.log 2>&1; f=$(find $O/prof_cb -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_cbam_$1x$2x$2_kernel_stats.csv; rm -rf $O/prof_cb; done
# phase stamps inside CBAM's F1 / B2 (or F1L / B2L) per stage: tools/bin/cbam_phase_probe is built in the container (hipcc, see its header)
if [ -x $R/tools/bin/cbam_phase_probe ]; then for sh in "64 28" "128 14" "256 7" "512 4"; do echo "== $sh"; timeout 60 $R/tools/bin/cbam_phase_probe $sh; done > $O/${T}_cbam_phases.txt 2>&1; fi
# round 5: where the scans sit in an UNTRACED step (HIP events around every scan launch), the GEMM tile kernels' counters, the chunk schedule's A/B
cd $R
M3T_BENCH_SCAN_TIMELINE=1 M3T_BENCH_EVENTS=1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --aux "" 2>&1 >/dev/null | grep "# tl" > $O/${T}_scan_timeline_untraced.txt
bash tools/gemm_pmc.sh 0 1 9600 1536 1024 > $O/${T}_gemm_pmc_nt_9600x1536x1024.txt 2>&1
REPS=2 STEPS=20 bash tools/ab.sh chunks "M3T_SCAN_CHUNKS=0" "M3T_SCAN_CHUNKS=fwd" "M3T_SCAN_CHUNKS=bwd" "M3T_SCAN_CHUNKS=1" "M3T_SCAN_FIRST=1" "M3T_SCAN_PREP_AHEAD=0 M3T_WEIGHT_AMAX=0" "M3T_GEMM_X6W=0" > $O/${T}_ab_switches.txt 2>&1
M3T_SCAN_CHUNKS=1 bash tools/tl.sh ${T}_chunks 60e3
ls $O | head -80
