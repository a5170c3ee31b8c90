#!/bin/bash
# Round-3 follow-up 2: deterministic ft_sumsq (ABI v6), two ranks on one GPU over gloo, waves-per-workgroup of attn_dqdk_k (A/B).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_i"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_optim.py -m gpu -x -q -k "attention or sumsq or radam or optim or guard or persistent" --timeout 400 --durations=5 -p no:cacheprovider > "$OUT/pytest_a.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_a.log"
for i in 1 2; do
FLOWTRON_TEST_SHARED_GPU=1 timeout 400 python -m pytest tests/test_gpu_dist.py -m gpu -q -s -k "two_ranks_on_one_gpu" --timeout 380 -p no:cacheprovider > "$OUT/pytest_shared_gpu_$i.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_shared_gpu_$i.log"
done
for G in 0 4 0 4; do
  FT_ATTN_DQDK_NW=$G timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > "$OUT/bench_nw$G.json" 2> "$OUT/bench_nw$G.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_nw$G.json')); print('NW=$G', d['ms_per_step'], d['value'])" >> "$OUT/ab.log" 2>&1
done
cd /tmp
rm -rf /tmp/kt && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > "$OUT/bench_under_rocprof.log" 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
cd "$REPO"
tail -n 12 "$OUT/pytest_a.log"
tail -n 8 "$OUT/pytest_shared_gpu_1.log"; tail -n 8 "$OUT/pytest_shared_gpu_2.log"
cat "$OUT/ab.log"
grep "attn_\|sumsq" "$OUT/bench_kernel_stats.csv" | cut -c1-60,150-260
