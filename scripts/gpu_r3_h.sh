#!/bin/bash
# Round-3 follow-up run: grouped dQ/dK kernel (tests + A/B), the T = 862 parity test on the faster padded-LSTM oracle,
# two ranks on one GPU over gloo (opt-in test), default bench line with the padded-LSTM cpu baseline.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_h"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_path.py -m gpu -x -q -k "attention or benchmark_config" --timeout 500 --durations=6 -p no:cacheprovider > "$OUT/pytest_a.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_a.log"
FLOWTRON_TEST_SHARED_GPU=1 timeout 400 python -m pytest tests/test_gpu_dist.py -m gpu -q -s -k "two_ranks_on_one_gpu" --timeout 380 -p no:cacheprovider > "$OUT/pytest_shared_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_shared_gpu.log"
for G in 1 0 1 0; do
  FT_ATTN_DQDK_GROUP=$G timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > "$OUT/bench_grp$G.json" 2> "$OUT/bench_grp$G.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_grp$G.json')); print('GROUP=$G', d['ms_per_step'], d['value'])" >> "$OUT/ab.log" 2>&1
done
cd /tmp
rm -rf /tmp/kt && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > "$OUT/bench_under_rocprof.log" 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
cd "$REPO"
timeout 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
tail -n 14 "$OUT/pytest_a.log"
tail -n 25 "$OUT/pytest_shared_gpu.log"
cat "$OUT/ab.log"
grep "attn_" "$OUT/bench_kernel_stats.csv" | cut -c1-60,150-260
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline'])"
