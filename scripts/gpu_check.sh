#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, a short bench; logs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c 'import torch;print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())' > gpurun_out/device.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -n 40 gpurun_out/pytest_gpu.log
tail -n 5 gpurun_out/smoke.log gpurun_out/bench.log
