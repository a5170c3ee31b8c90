#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider -k "train or loop or stft or infer" > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_sel.log
tail -n 40 gpurun_out/pytest_sel.log
timeout 120 python scripts/prof_stft.py 2>&1 | tail -2
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['infer']))"
