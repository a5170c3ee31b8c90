#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_i"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q --timeout 600 -p no:cacheprovider -s -k "decode_400 or infer_vs_reference or infer_bf16 or depths" > "$OUT/pytest_sel.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_sel.log"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-trainpy > "$OUT/bench_line.json" 2> "$OUT/bench.err"
grep "decode 400\|mel max\|passed\|failed\|Error" "$OUT/pytest_sel.log" | tail -12
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step']); print(json.dumps(d['infer'],indent=0)[:400]); print(json.dumps(d['infer_fp32'],indent=0)[:900])"
