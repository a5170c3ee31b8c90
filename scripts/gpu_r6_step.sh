#!/bin/bash
# round 6: the step with the decoder layer pair pipeline (FLOWTRON_LSTM_PAIR / _PAIR_BWD) -- A/B on one box + targeted parity tests
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r6_${1:-step}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for cfg in ${CFGS:-"FLOWTRON_LSTM_PAIR=6" "FLOWTRON_LSTM_PAIR=0" "FLOWTRON_LSTM_PAIR=0,FLOWTRON_LSTM_ROLES=0" "FLOWTRON_LSTM_PAIR=6" "FLOWTRON_LSTM_PAIR=5" "FLOWTRON_LSTM_PAIR=4" "FLOWTRON_LSTM_PAIR=6,FLOWTRON_LSTM_PAIR_BWD=3" "FLOWTRON_LSTM_PAIR=6,FLOWTRON_LSTM_PAIR_BWD=4"}; do
    echo "$cfg: $(env ${cfg//,/ } timeout 300 python bench.py --steps 30 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep -E 'timed region|Error|error' | tail -n 2)" | tee -a "$OUT/pair_ab.log"
done
timeout 1500 python -m pytest ${TESTS:-tests/test_gpu_bench_path.py tests/test_gpu_model.py} -m gpu -q --timeout 900 -p no:cacheprovider -x > "$OUT/pytest_gpu_part.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu_part.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu_part.log" | grep -E "passed|failed|FAILED|ERROR|Error|assert" | tail -n 12
