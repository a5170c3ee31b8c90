#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -k "big_tile or gemm or linear or compact" -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_f.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_f.log
timeout 200 python scripts/exp/gemm_img_bench.py > gpurun_out/gemm_img_bench.log 2>&1
for BIG in 1 0; do
  FT_GEMM_BF16_BIG=$BIG timeout 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-infer > gpurun_out/bench_f_big$BIG.json 2> gpurun_out/bench_f_big$BIG.err
done
tail -n 12 gpurun_out/pytest_f.log
grep -v amdgpu.ids gpurun_out/gemm_img_bench.log
for f in gpurun_out/bench_f_big*.json; do echo $f; head -c 250 $f; echo; done
tail -n 3 gpurun_out/bench_f_big1.err
