#!/bin/bash
# Round-3 run B (GPU box, via gpurun): compact GEMMs + attention changes -- op / model / bench-path / fp16 tests, the default bench
# line, the cumulative-attention bench line, and a kernel-trace of three training steps.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bench_path.py tests/test_gpu_fp16.py -m gpu -q --timeout 900 -p no:cacheprovider -s > gpurun_out/pytest_b.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_b.log
timeout 600 python bench.py > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
echo "bench exit $?" >> gpurun_out/bench_b.err
FLOWTRON_LSTM_PERSIST=11 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-infer > gpurun_out/bench_b_bare.json 2> gpurun_out/bench_b_bare.err
echo "bench bare exit $?" >> gpurun_out/bench_b_bare.err
timeout 300 python scripts/exp/lstm_persist_bench.py > gpurun_out/persist_bench_b.log 2>&1
timeout 400 python bench.py --config ljs_cumm --steps 2 --warmup 1 --no-infer --no-cpu-baseline > gpurun_out/bench_b_cumm.json 2> gpurun_out/bench_b_cumm.err
echo "bench cumm exit $?" >> gpurun_out/bench_b_cumm.err
cd /tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer > $REPO/gpurun_out/bench_b_rocprof.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/bench_b_kernel_stats.csv 2>/dev/null
cd "$REPO"
grep -E "passed|failed|rel-L2|decode 400|mel max|cumulative attention|FAILED|Error" gpurun_out/pytest_b.log | head -60
tail -n 3 gpurun_out/bench_b.err gpurun_out/bench_b_cumm.err
head -c 1200 gpurun_out/bench_b.json; echo
head -c 400 gpurun_out/bench_b_bare.json; echo
grep -v amdgpu.ids gpurun_out/persist_bench_b.log | tail -n 30
head -c 900 gpurun_out/bench_b_cumm.json; echo
head -n 25 gpurun_out/bench_b_kernel_stats.csv | cut -c1-160
