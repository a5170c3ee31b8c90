#!/bin/bash
# deterministic split-K check: unit test, forward noise, step time in the three encoder split-K modes, the DP tests
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-det}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider \
   -k "deterministic_split or dist or rccl or ranks or wide_batch or conv_norm or golden" --deselect tests/test_gpu_model.py::test_decode_400_frames_vs_oracle_all_three_decoders > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 8
python scripts/exp/noise_debug.py big 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|return {" | head -3
for m in det 0 atomic; do
  for k in 1 2; do
    echo "ENC_SPLITK=$m: $(FLOWTRON_ENC_SPLITK=$m timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/enc_splitk_sweep.log"
  done
done
