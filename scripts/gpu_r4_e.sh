#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_e"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 500 -p no:cacheprovider -k "reduce_scatter" > "$OUT/pytest_rs.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_rs.log"
timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-infer > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -n 4 "$OUT/pytest_rs.log"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d.get('trainpy_step'), indent=1))"
