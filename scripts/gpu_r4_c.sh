#!/bin/bash
# Round-4 call C: where the train.py call sequence loses time against the headline step (kernel timeline of a trainpy step) + DP tests
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_c"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q --timeout 500 -p no:cacheprovider > "$OUT/pytest_dist.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_dist.log"
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer > "$OUT/bench_under_rocprof.log" 2>&1
TR=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/trainpy_step_timeline.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
cd "$REPO"
tail -n 6 "$OUT/pytest_dist.log"
head -n 75 "$OUT/trainpy_step_timeline.txt"
