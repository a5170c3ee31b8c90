#!/bin/bash
# Round-3 run E: L2-aware tile order of the image GEMM -- correctness (GEMM / linear / compact tests, bench-path parity) and A/B benches.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -k "gemm or linear or compact or image or lstm_seq" tests/test_gpu_fp16.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_e.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_e.log
for V in "1 0" "0 0" "1 256"; do
  set -- $V
  FT_GEMM_BF16_ORDER=$1 FT_GEMM_BF16_TILE=$2 timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > gpurun_out/bench_e_order$1_tile$2.json 2> gpurun_out/bench_e_order$1_tile$2.err
done
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer > $REPO/gpurun_out/bench_e_rocprof.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/bench_e_kernel_stats.csv 2>/dev/null
cd "$REPO"
tail -n 5 gpurun_out/pytest_e.log
for f in gpurun_out/bench_e_order*.json; do echo $f; head -c 260 $f; echo; done
grep gemm gpurun_out/bench_e_kernel_stats.csv | cut -c1-160
