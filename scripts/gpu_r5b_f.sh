#!/bin/bash
# non-temporal C stores of the large forward GEMM outputs (FT_GEMM_NT = MB threshold) + the fixed shared-activation test
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-f}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -k "shared_activation or test_gemm" > "$OUT/pytest_gpu.log" 2>&1
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 5
for m in 0 64 0 64 16; do
    echo "GEMM_NT=$m: $(FT_GEMM_NT=$m timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/gemm_nt_sweep.log"
done
