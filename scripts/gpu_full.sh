#!/bin/bash
# full GPU check: the -m gpu suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -v "Warning\|warn\|^$\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/pytest_gpu.log | tail -n 25
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
