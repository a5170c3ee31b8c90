#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_k"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ops.py -m gpu -x -q -k "dist or attention" --timeout 500 --durations=8 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -n 25 "$OUT/pytest.log"
