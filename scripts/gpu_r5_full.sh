#!/bin/bash
# Round 5: full -m gpu suite, smoke(), default bench line.  usage: gpu_r5_full.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-full}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 --durations=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | tail -n 16
tail -n 3 "$OUT/smoke.log"
tail -n 3 "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline']['value'], d['infer']['rtf'], d['infer_fp32']['rtf'], d['parity']); print(d.get('trainpy_step')); r=d['roofline']; print(r['dominant_kernel']['kernel'], r['dominant_kernel']['frac'], r['dominant_kernel'].get('floor_frac'), r['second_kernel']['kernel'], r['second_kernel']['frac'])"
