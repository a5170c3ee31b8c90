#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_s"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 300 python scripts/exp/bilstm_persist_check.py 2>&1 | grep -v amdgpu.ids > "$OUT/check.log"
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "persistent_bilstm or golden" --timeout 200 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
cat "$OUT/check.log"; tail -n 5 "$OUT/pytest.log"
