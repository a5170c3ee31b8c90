#!/bin/bash
# Round-3 evidence run (GPU box, via gpurun): full -m gpu suite, smoke(), the default bench line, kernel-trace stats of
# the bench command and the PMC passes over the training step (one counter group per rocprofv3 run, kernel dispatch
# tracing only).  Every stage has its own timeout; summaries land in gpurun_out/r3_final/.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_final"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 --durations=15 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 300 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
cd /tmp
rm -rf /tmp/kt && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/bench_under_rocprof.log"
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  TAG=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_run
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- env BENCH_INFER_FRAMES=64 BENCH_INFER_CALLS=1 python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${TAG}_bench.log" 2>&1
  echo "rocprof pmc [$C] exit $?" >> "$OUT/pmc_${TAG}_bench.log"
  F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/pmc_${TAG}_bench.json" > /dev/null
done
cd "$REPO"
timeout 200 python bench.py --config libritts --steps 30 --warmup 3 --no-cpu-baseline --no-infer > "$OUT/bench_line_libritts.json" 2> "$OUT/bench_libritts.err"
ls -la "$OUT"
tail -n 30 "$OUT/pytest_gpu.log"
cat "$OUT/smoke.log" | tail -3
head -c 600 "$OUT/bench_line.json"; echo
head -n 14 "$OUT/bench_kernel_stats.csv"
