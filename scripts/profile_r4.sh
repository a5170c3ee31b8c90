#!/bin/bash
# Round-4 evidence run (GPU box, via gpurun): kernel-trace stats + per-instance table + ordered step timeline of the training step, PMC
# passes (HBM traffic, MFMA-busy, wave cycles; one counter group per rocprofv3 run, kernel dispatch tracing only) over the bench
# workload and the STFT front end, the recurrence micro-benchmark and the probes.  Summaries land in gpurun_out/prof_r4/; the ones
# to be judged are copied to profiles/r04_* by hand.   usage: profile_r4.sh [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_r4${1:+_$1}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command (training steps + roofline legs + inference), and of training steps alone
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/bench_under_rocprof.log"
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/train_under_rocprof.log" 2>&1
TR=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
python $REPO/scripts/kernel_trace_table.py "$TR" 30 > "$OUT/kernel_instances.txt" 2>&1
# 2. PMC passes
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  TAG=$(echo $C | cut -d' ' -f1)
  for WL in bench stft; do
    if [ $WL = bench ]; then CMD="env BENCH_INFER_FRAMES=64 BENCH_INFER_CALLS=1 python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-trainpy"; else CMD="python $REPO/scripts/prof_stft.py"; fi
    rm -rf /tmp/pmc_run
    timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- $CMD > "$OUT/pmc_${TAG}_${WL}.log" 2>&1
    echo "rocprof pmc [$C] $WL exit $?" >> "$OUT/pmc_${TAG}_${WL}.log"
    F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/pmc_${TAG}_${WL}.json" > /dev/null
  done
done
# 3. recurrence micro-benchmark (us per step, phase stamps) and the probes
cd "$REPO"
FWD_NGS=1,11 BWD_NGS=1,11,21 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench.log" 2>&1
timeout 60 ./scripts/exp/mfma_rate_probe > "$OUT/mfma_rate_probe.log" 2>&1
timeout 60 ./scripts/exp/cumask_probe > "$OUT/cumask_probe.log" 2>&1
ls -la "$OUT"
head -n 14 "$OUT/bench_kernel_stats.csv" | cut -c1-160
head -n 12 "$OUT/step_timeline.txt"
grep "persistent" "$OUT/persist_bench.log"
