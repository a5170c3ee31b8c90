#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_p"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "persistent_lstm_forward" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
tail -n 3 "$OUT/pytest.log"
FWD_NGS=1,31 BWD_NGS=21 PROF_NG=31 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench.log" 2>&1
grep -v amdgpu.ids "$OUT/persist_bench.log" | head -12
timeout 60 ./scripts/exp/mfma_rate_probe > "$OUT/mfma_rate_probe.log" 2>&1
FLOWTRON_LSTM_PERSIST_FWD=ms timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_ms.json" 2> "$OUT/bench_ms.err"
python -c "
import json; d=json.load(open('$OUT/bench_line_ms.json')); print('step with the M-split forward:', d['ms_per_step'], d['value'])"
