#!/bin/bash
# Round-4 call A: CU-mask probe + ordered kernel timeline of ONE training step of the round-3 code (where does the GPU idle?).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_a"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 60 ./scripts/exp/cumask_probe > "$OUT/cumask_probe.log" 2>&1
echo "probe exit $?" >> "$OUT/cumask_probe.log"
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer > "$OUT/bench_under_rocprof.log" 2>&1
TR=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
python $REPO/scripts/kernel_trace_table.py "$TR" 30 > "$OUT/kernel_instances.txt" 2>&1
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
cd "$REPO"
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-infer > "$OUT/bench_line_train_only.json" 2> "$OUT/bench.err"
cat "$OUT/cumask_probe.log"
head -n 8 "$OUT/step_timeline.txt"
python -c "import json; d=json.load(open('$OUT/bench_line_train_only.json')); print(d['ms_per_step'], d['value'])"
