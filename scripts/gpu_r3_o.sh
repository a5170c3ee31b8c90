#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_o"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_path.py tests/test_gpu_train_loop.py -m gpu -x -q -k "dgates_image or image_only or benchmark_config or libritts or lstm_layer or full_width_loop" --timeout 500 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
for G in 1 0 1 0; do
  FLOWTRON_LSTM_PERSIST_IMG=$G timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > "$OUT/bench_img$G.json" 2> "$OUT/bench_img$G.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_img$G.json')); print('PERSIST_IMG=$G', d['ms_per_step'], d['value'])" >> "$OUT/ab.log" 2>&1
done
tail -n 12 "$OUT/pytest.log"; cat "$OUT/ab.log"
