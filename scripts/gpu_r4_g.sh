#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_g"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 500 -p no:cacheprovider -k "persistent or reduce_scatter or concatenated or two_inputs" > "$OUT/pytest_sel.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_sel.log"
FWD_NGS=1,11 BWD_NGS=11,21 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench.log" 2>&1
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_train_only.json" 2> "$OUT/bench.err"
tail -n 5 "$OUT/pytest_sel.log"
grep "persistent\|^wave\|bwd wave" "$OUT/persist_bench.log"
python -c "import json; d=json.load(open('$OUT/bench_line_train_only.json')); print(d['ms_per_step'], d['value'])"
