#!/bin/bash
# Round-5 evidence run (GPU box, via gpurun): kernel-trace stats + ordered step timeline of the training step, the ljs_cumm line's
# kernel stats, PMC passes (HBM traffic, MFMA-busy, wave cycles; one counter group per rocprofv3 run, kernel dispatch tracing only)
# over the bench workload and over the fused cumulative-attention frames, the stage stamps.  Summaries land in gpurun_out/prof_r5/;
# the ones to be judged are copied to profiles/r05_* by hand.   usage: profile_r5.sh [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_r5${1:+_$1}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/bench_under_rocprof.log"
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/train_under_rocprof.log" 2>&1
TR=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
python $REPO/scripts/kernel_trace_table.py "$TR" 30 > "$OUT/kernel_instances.txt" 2>&1
rm -rf /tmp/kt3 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o cumm -- python $REPO/bench.py --config ljs_cumm --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/cumm_under_rocprof.log" 2>&1
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) "$OUT/ljs_cumm_kernel_stats.csv" 2>/dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  TAG=$(echo $C | cut -d' ' -f1)
  for WL in bench cumm; do
    if [ $WL = bench ]; then CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-trainpy --no-infer"; else CMD="python $REPO/scripts/exp/cumm_prof.py 24"; fi
    rm -rf /tmp/pmc_run
    timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- $CMD > "$OUT/pmc_${TAG}_${WL}.log" 2>&1
    echo "rocprof pmc [$C] $WL exit $?" >> "$OUT/pmc_${TAG}_${WL}.log"
    F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/pmc_${TAG}_${WL}.json" > /dev/null
  done
done
cd "$REPO"
timeout 120 python scripts/exp/cumm_prof.py 60 > "$OUT/cumm_stage_stamps.log" 2>&1
ls -la "$OUT"
head -n 12 "$OUT/bench_kernel_stats.csv" | cut -c1-160
head -n 14 "$OUT/step_timeline.txt"
head -n 6 "$OUT/ljs_cumm_kernel_stats.csv" | cut -c1-160
tail -n 4 "$OUT/cumm_stage_stamps.log"
