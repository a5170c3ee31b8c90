#!/bin/bash
# wide batch up to 128: kernel-level test + step time at B = 96 / 128 with and without the sliced persistent launches
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-i}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -k "wide_batch" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error" | tail -4
for bm in "96 1" "96 0" "128 1" "128 0"; do set -- $bm
    echo "B=$1 PERSIST_WIDE=$2: $(FLOWTRON_LSTM_PERSIST_WIDE=$2 timeout 300 python bench.py --batch $1 --steps 8 --warmup 2 --no-infer --no-trainpy --no-cpu-baseline 2>&1 | grep -E 'timed region|Error|error' | tail -2)" | tee -a "$OUT/wide_batch.log"
done
