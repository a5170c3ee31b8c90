#!/bin/bash
# Round 5: fused cumulative attention -- op-level parity, model-level goldens, the ljs_cumm bench line.  usage: gpu_r5_cumm.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-cumm}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s --timeout 300 -p no:cacheprovider -k "fused_cumulative" > "$OUT/pytest_ops.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_ops.log"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "cumm or cumulative" > "$OUT/pytest_model.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_model.log"
timeout 600 python bench.py --config ljs_cumm --steps 3 --warmup 1 --no-infer --no-trainpy --no-cpu-baseline > "$OUT/bench_line_ljs_cumm.json" 2> "$OUT/bench_cumm.err"
echo "bench exit $?" >> "$OUT/bench_cumm.err"
tail -n 30 "$OUT/pytest_ops.log"
tail -n 15 "$OUT/pytest_model.log"
tail -n 5 "$OUT/bench_cumm.err"
python -c "
import json; d=json.load(open('$OUT/bench_line_ljs_cumm.json')); print('ljs_cumm ms/step', d['ms_per_step'], d['value'])"
