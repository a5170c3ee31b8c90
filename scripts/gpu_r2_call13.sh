#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -n 4 gpurun_out/smoke.log gpurun_out/bench.log
FLOWTRON_LSTM_PERSIST=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-infer > gpurun_out/bench_nopersist.log 2>&1
tail -n 2 gpurun_out/bench_nopersist.log
bash scripts/profile_r2.sh > gpurun_out/profile_r2.log 2>&1
tail -n 40 gpurun_out/profile_r2.log
