#!/bin/bash
# Round 5 (second session): attention forward tile height A/B (FLOWTRON_ATTN_TT) on the default step + the tests the first check did not reach
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-attn}"
OUT="$REPO/gpurun_out/r5b_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for tt in 32 16 82 8; do
  for k in 1 2; do
    echo "ATTN_TT=$tt: $(FLOWTRON_ATTN_TT=$tt timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/attn_tt_sweep.log"
  done
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_optim.py tests/test_gpu_dist.py \
    -m gpu -q --timeout 600 --durations=6 -p no:cacheprovider \
    -k "test_attention or wide_batch or golden or optim or radam or dist or rccl or ranks" \
    --deselect tests/test_gpu_model.py::test_decode_400_frames_vs_oracle_all_three_decoders > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 12
