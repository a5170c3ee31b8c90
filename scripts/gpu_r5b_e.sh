#!/bin/bash
# in-place dx accumulation + _acc column sums: ops / model tests, then the step time
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-e}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bench_path.py -m gpu -q --timeout 600 -p no:cacheprovider \
   --deselect tests/test_gpu_model.py::test_decode_400_frames_vs_oracle_all_three_decoders -k "not persistent_lstm and not decode" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 8
for m in 1 0; do
  for k in 1 2; do
    echo "DX_INPLACE=$m: $(FLOWTRON_DX_INPLACE=$m timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')"
  done
done
