#!/bin/bash
# batches wider than the recurrence kernels take: model-level test, kernel-level test, step time at B = 96 / 128
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-i}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider -k "wide_batch or wider_than or unsupported_sizes" > "$OUT/pytest_gpu.log" 2>&1
grep -v amdgpu.ids "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|Error|assert" | tail -8
for b in 96 128; do
    echo "B=$b: $(timeout 300 python bench.py --batch $b --steps 8 --warmup 2 --no-infer --no-trainpy --no-cpu-baseline 2>&1 | grep -E 'timed region|Error|error' | tail -2)" | tee -a "$OUT/wide_batch.log"
done
