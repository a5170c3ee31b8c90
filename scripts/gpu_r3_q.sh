#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_q"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bench_path.py -m gpu -x -q -k "persistent_bilstm or golden or benchmark_config_vs or libritts or encoder or bilstm" --timeout 300 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
for G in 1 0 1 0; do
  FLOWTRON_BILSTM_PERSIST=$G timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > "$OUT/bench_bp$G.json" 2> "$OUT/bench_bp$G.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_bp$G.json')); print('BILSTM_PERSIST=$G', d['ms_per_step'], d['value'])" >> "$OUT/ab.log" 2>&1
done
tail -n 8 "$OUT/pytest.log"; cat "$OUT/ab.log"
