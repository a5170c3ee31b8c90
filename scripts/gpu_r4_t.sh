#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_t"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "persistent or compact_lstm" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
tail -n 3 "$OUT/pytest.log"
FWD_NGS=1,11 BWD_NGS=21 PROF_NG=11 PROF_NG_BWD=21 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench.log" 2>&1
grep "persistent ng\|^wave [02]" "$OUT/persist_bench.log"
timeout 400 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print('step:', d['ms_per_step'], d['value'])"
