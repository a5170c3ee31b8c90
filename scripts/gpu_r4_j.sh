#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_j"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gate_layer or concatenated or gemm" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
tail -n 25 "$OUT/pytest.log"
timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -n 3 "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'])"
FLOWTRON_GATE_ON_CAT=0 timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_off.json" 2> "$OUT/bench_off.err"
python -c "
import json; d=json.load(open('$OUT/bench_line_off.json')); print('gate fusion off', d['ms_per_step'], d['value'])"
