#!/bin/bash
# wide batch (B = 40 / 48) end to end: correctness against the launch-per-step kernels, step time with / without the sliced persistent launches
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-h}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
python scripts/exp/wide_batch_check.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -3 | tee "$OUT/wide_batch.log"
for m in 1 0; do
    echo "B=48 PERSIST_WIDE=$m: $(FLOWTRON_LSTM_PERSIST_WIDE=$m timeout 300 python bench.py --batch 48 --steps 20 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 | grep -E 'timed region|Error|error' | tail -2)" | tee -a "$OUT/wide_batch.log"
done
