#!/bin/bash
# Round-4 call H: optimizer / dist / train-loop tests after the lazy zero_grad, secondary configs (libritts bf16 + fp16) with kernel stats
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_h"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_optim.py tests/test_gpu_dist.py tests/test_gpu_train_loop.py tests/test_gpu_fp16.py -m gpu -x -q --timeout 500 -p no:cacheprovider > "$OUT/pytest_sel.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_sel.log"
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_ljs.json" 2> "$OUT/bench_ljs.err"
timeout 200 python bench.py --config libritts --steps 30 --warmup 5 --no-trainpy > "$OUT/bench_line_libritts.json" 2> "$OUT/bench_libritts.err"
timeout 200 python bench.py --config libritts_fp16 --steps 30 --warmup 5 --no-trainpy > "$OUT/bench_line_libritts_fp16.json" 2> "$OUT/bench_libritts_fp16.err"
cd /tmp
for C in libritts libritts_fp16; do
  rm -rf /tmp/kt_$C && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$C -o bench -- python $REPO/bench.py --config $C --steps 3 --warmup 2 --no-trainpy > "$OUT/rocprof_$C.log" 2>&1
  TR=$(find /tmp/kt_$C -name "*kernel_trace.csv" | head -1)
  python $REPO/scripts/step_timeline.py "$TR" 1 | head -50 > "$OUT/step_timeline_$C.txt" 2>&1
  cp $(find /tmp/kt_$C -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_$C.csv" 2>/dev/null
done
cd "$REPO"
tail -n 5 "$OUT/pytest_sel.log"
for C in ljs libritts libritts_fp16; do python -c "import json; d=json.load(open('$OUT/bench_line_$C.json')); print('$C', d['ms_per_step'], d['value'], d['config'].get('skipped_steps'))"; done
head -30 "$OUT/step_timeline_libritts_fp16.txt"
