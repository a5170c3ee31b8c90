#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_q"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "compact" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
tail -n 12 "$OUT/pytest.log"
