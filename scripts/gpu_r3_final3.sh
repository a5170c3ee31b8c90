#!/bin/bash
# Round-3 closing confirmation of the final state (ABI v8): full -m gpu suite, smoke(), default bench line, kernel-trace stats.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_final3"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 --durations=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
cd /tmp
rm -rf /tmp/kt && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
python $REPO/scripts/kernel_trace_table.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) 100 > "$OUT/kernel_instances.txt" 2>&1
cd "$REPO"
tail -n 14 "$OUT/pytest_gpu.log"
tail -n 2 "$OUT/smoke.log"
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline']['value'], d['infer']['rtf'], d['infer_fp32']['rtf'], d['parity']['worst_grad_rel_well_conditioned'])"
head -n 12 "$OUT/bench_kernel_stats.csv" | cut -c1-150
