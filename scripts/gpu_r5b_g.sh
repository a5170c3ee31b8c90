#!/bin/bash
# device-side GradScaler skip (RAdam._step_supports_amp_scaling + ft_radam_step_dev): optimizer / fp16 / loop tests, then the fp16 config A/B
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-g}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_fp16.py tests/test_gpu_train_loop.py -m gpu -q --timeout 600 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 6
for m in 0 1 0 1; do
    echo "AMP_HOST_SKIP=$m: $(FLOWTRON_AMP_HOST_SKIP=$m timeout 300 python bench.py --config libritts_fp16 --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/amp_skip_sweep.log"
done
echo "bf16: $(timeout 300 python bench.py --config libritts --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/amp_skip_sweep.log"
