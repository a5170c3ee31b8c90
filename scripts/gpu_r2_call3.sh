#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python scripts/exp/lstm_persist_bench.py > gpurun_out/persist_bench.log 2>&1
echo "persist bench exit $?" >> gpurun_out/persist_bench.log
cat gpurun_out/persist_bench.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_optim.py -m gpu -q --timeout 300 -p no:cacheprovider -k "persistent or radam or lstm or optim or resume or gradient" > gpurun_out/pytest_gpu_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_sel.log
tail -n 30 gpurun_out/pytest_gpu_sel.log
