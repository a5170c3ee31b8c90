#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/exp/decode_prof.py > gpurun_out/decode_prof.log 2>&1
cat gpurun_out/decode_prof.log | tail -15
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider -k "infer" > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_sel.log
tail -n 8 gpurun_out/pytest_sel.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['infer']))"
