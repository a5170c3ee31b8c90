#!/bin/bash
# Round 5 (second session) closing run: full -m gpu suite, smoke(), the default bench line, the secondary configurations, and the
# kernel-trace stats + ordered step timeline of the same code.  usage: gpu_r5b_final.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-final}"
OUT="$REPO/gpurun_out/r5b_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
timeout 300 python bench.py --config ljs_cumm --steps 5 --warmup 2 --no-infer --no-trainpy --no-cpu-baseline > "$OUT/bench_line_ljs_cumm.json" 2> "$OUT/bench_cumm.err"
timeout 300 python bench.py --config libritts --steps 40 --no-infer --no-cpu-baseline > "$OUT/bench_line_libritts.json" 2> "$OUT/bench_libritts.err"
timeout 300 python bench.py --config libritts_fp16 --steps 40 --no-infer --no-cpu-baseline > "$OUT/bench_line_libritts_fp16.json" 2> "$OUT/bench_libritts_fp16.err"
bash scripts/profile_r5b.sh "$TAG" > "$OUT/profile.log" 2>&1
cp "$REPO/gpurun_out/prof_r5b_$TAG/step_timeline.txt" "$REPO/gpurun_out/prof_r5b_$TAG/train_kernel_stats.csv" "$REPO/gpurun_out/prof_r5b_$TAG/kernel_instances.txt" "$OUT/" 2>/dev/null
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR" | tail -n 8
tail -n 2 "$OUT/smoke.log"
tail -n 2 "$OUT/bench.err"
head -n 2 "$OUT/step_timeline.txt"
python -c "
import json
d=json.load(open('$OUT/bench_line.json')); print('ljs', d['ms_per_step'], d['value'], 'cpu', d['cpu_baseline']['value'], 'rtf', d['infer']['rtf'], d['infer_fp32']['rtf'], 'parity', d['parity']['worst_grad_rel'], 'trainpy', d.get('trainpy_step',{}).get('ms_per_step'), 'gap', d.get('trainpy_step',{}).get('gap_to_headline_ms'))
r=d['roofline']; print(r['dominant_kernel']['kernel'], r['dominant_kernel']['frac'], r['dominant_kernel'].get('floor_frac'), r['dominant_kernel'].get('share_of_step'), '|', r['second_kernel']['kernel'], r['second_kernel']['frac'], r['second_kernel'].get('share_of_step'))
for n in ('ljs_cumm','libritts','libritts_fp16'):
    try:
        d=json.load(open('$OUT/bench_line_%s.json' % n)); print(n, d['ms_per_step'], d['value'])
    except Exception as e: print(n, 'ERR', e)
"
