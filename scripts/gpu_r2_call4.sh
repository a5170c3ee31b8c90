#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python scripts/exp/lstm_persist_bench.py > gpurun_out/persist_bench.log 2>&1
echo "persist bench exit $?" >> gpurun_out/persist_bench.log
cat gpurun_out/persist_bench.log
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -p no:cacheprovider -k "persistent" 2>&1 | tail -5
