#!/bin/bash
# round-2 GPU call 1: persistent-LSTM microbench first (bounded), then the whole -m gpu suite, smoke, a short bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c 'import torch;print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info(), torch.cuda.get_device_properties(0).multi_processor_count)' > gpurun_out/device.log 2>&1
timeout 180 python scripts/exp/lstm_persist_bench.py > gpurun_out/persist_bench.log 2>&1
echo "persist bench exit $?" >> gpurun_out/persist_bench.log
cat gpurun_out/persist_bench.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -n 5 gpurun_out/smoke.log gpurun_out/bench.log
