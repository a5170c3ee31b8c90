#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_u"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for F in ksplit bare ksplit bare; do
FLOWTRON_LSTM_PERSIST_FWD=$F timeout 400 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_$F.json" 2> "$OUT/bench_$F.err"
python -c "
import json; d=json.load(open('$OUT/bench_$F.json')); print('$F', d['ms_per_step'], d['value'], d['roofline']['second_kernel']['us_per_step'])"
done
