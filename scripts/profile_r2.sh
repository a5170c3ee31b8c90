#!/bin/bash
# Round-2 evidence run (GPU box, via gpurun): kernel-trace stats of the bench command + PMC passes (HBM traffic and
# MFMA-busy counters) over the bench workload (training step, roofline microbenchmarks, inference) and the STFT front end.
# One counter group per rocprofv3 run, kernel dispatch tracing only (MI355X_MICROARCH.md: TCC has 4 slots -- FETCH_SIZE
# takes 3, WRITE_SIZE 2; never combined with sys/hip/hsa trace domains).  Summaries land in gpurun_out/prof_r2/.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_r2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps ${PROF_STEPS:-2} --warmup 1 --no-cpu-baseline"
# 1. kernel trace + stats
rm -rf /tmp/kt && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/bench_under_rocprof.log"
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
# 2. PMC passes
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  TAG=$(echo $C | cut -d' ' -f1)
  for WL in bench stft; do
    if [ $WL = bench ]; then CMD="env BENCH_INFER_FRAMES=64 BENCH_INFER_CALLS=1 python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline"; else CMD="python $REPO/scripts/prof_stft.py"; fi
    rm -rf /tmp/pmc_run
    timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- $CMD > "$OUT/pmc_${TAG}_${WL}.log" 2>&1
    echo "rocprof pmc [$C] $WL exit $?" >> "$OUT/pmc_${TAG}_${WL}.log"
    F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/pmc_${TAG}_${WL}.json" > /dev/null
  done
done
ls -la "$OUT"
head -n 30 "$OUT/bench_kernel_stats.csv"
