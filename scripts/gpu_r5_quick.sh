#!/bin/bash
# Round 5: quick check -- fused cumulative attention tests + stamps + ljs_cumm line, then the default line (30 steps, parity leg)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-quick}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
bash scripts/gpu_r5_cumm3.sh "$TAG" 2>&1 | tail -n 9
for k in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
done
timeout 400 python bench.py --steps 30 --warmup 3 --no-infer > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print('ljs', d['ms_per_step'], 'parity', d['parity']['worst_grad_rel'], d['parity']['worst_grad_name'], 'trainpy', d['trainpy_step']['ms_per_step'], d['trainpy_step']['gap_to_headline_ms'])"
