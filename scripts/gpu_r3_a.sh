#!/bin/bash
# Round-3 run A (GPU box, via gpurun): the new parity / survivability tests + the default bench line + persistent-kernel stamps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c 'import torch,os;print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info(), os.cpu_count()); import psutil; print(psutil.virtual_memory())' > gpurun_out/device.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_optim.py tests/test_gpu_dist.py "tests/test_gpu_model.py::test_decode_400_frames_vs_oracle_all_three_decoders" tests/test_gpu_ops.py -k "not trajectory" -m gpu -q --timeout 900 -p no:cacheprovider -s > gpurun_out/pytest_a.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_a.log
timeout 600 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
echo "bench exit $?" >> gpurun_out/bench_a.err
timeout 300 python scripts/exp/lstm_persist_bench.py > gpurun_out/persist_bench_a.log 2>&1
tail -n 60 gpurun_out/pytest_a.log
tail -n 3 gpurun_out/bench_a.err
cat gpurun_out/bench_a.json | head -c 3000
tail -n 20 gpurun_out/persist_bench_a.log
