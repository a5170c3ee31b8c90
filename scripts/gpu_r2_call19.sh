#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fp16.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_f16.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_f16.log
grep -v Warning gpurun_out/pytest_f16.log | tail -n 30
timeout 1500 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_ops.py tests/test_gpu_bench_path.py tests/test_gpu_optim.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_sel.log
tail -n 12 gpurun_out/pytest_sel.log
for m in bf16 f16; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mfma $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], d.get('dtype'), json.dumps(d.get('parity'))[:300])"
done
for c in libritts libritts_fp16; do
timeout 600 python bench.py --steps 3 --warmup 1 --config $c 2>gpurun_out/bench_$c.err | tee gpurun_out/bench_$c.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d.get('dtype'), d['config']['workload'][:150])" || tail -5 gpurun_out/bench_$c.err
done
