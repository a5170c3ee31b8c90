#!/bin/bash
# Round 5: bf16 parity of the bench line with the encoder in fp32 operands.  usage: gpu_r5_enc32.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-enc32}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for v in off split off split fwd; do
    FLOWTRON_ENCODER_F32=$v timeout 500 python bench.py --steps 30 --warmup 3 --no-infer --no-trainpy > "$OUT/bench_enc32_${v:-off}.json" 2> "$OUT/bench_enc32_${v:-off}.err"
    python -c "
import json; d=json.load(open('$OUT/bench_enc32_${v:-off}.json')); print('encoder_f32=%-5s' % '${v:-off}', d['ms_per_step'], 'ms/step | parity', json.dumps(d.get('parity'))[:600])" | tee -a "$OUT/enc32.log"
done
