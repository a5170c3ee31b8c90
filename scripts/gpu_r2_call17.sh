#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fp16.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/pytest_f16.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_f16.log
tail -n 40 gpurun_out/pytest_f16.log
timeout 900 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_ops.py tests/test_gpu_bench_path.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_sel.log
tail -n 12 gpurun_out/pytest_sel.log
