#!/bin/bash
# Round 5 (second session): targeted check of the fused loss / arena gradients / run-aware embedding backward / sliced persistent launches,
# then the default line without the decode and CPU legs.  usage: gpu_r5b_check.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-check}"
OUT="$REPO/gpurun_out/r5b_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_optim.py tests/test_gpu_train_loop.py tests/test_gpu_dist.py \
    -m gpu -q --timeout 600 --durations=6 -p no:cacheprovider -x \
    -k "fused_flowtron_loss or embedding or attention_ctc or wide_batch or affine_and_losses or golden or optim or radam or train or dist or rccl or ranks or bilstm" \
    --deselect tests/test_gpu_model.py::test_decode_400_frames_vs_oracle_all_three_decoders > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error|assert" | tail -n 12
for k in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"
done
timeout 400 python bench.py --steps 30 --warmup 3 --no-infer --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -n 3 "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print('ljs', d['ms_per_step'], 'trainpy', d['trainpy_step']['ms_per_step'], d['trainpy_step']['gap_to_headline_ms'], d['trainpy_step']['host_ms'])"
