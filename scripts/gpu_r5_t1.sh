#!/bin/bash
# Round 5: targeted GPU tests of this round's changes.  usage: gpu_r5_t1.sh <tag> <pytest -k expression> [files...]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-t1}"; shift
KEXPR="$1"; shift
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 1500 python -m pytest "$@" -m gpu -q -s --timeout 600 -p no:cacheprovider -k "$KEXPR" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
grep -v "amdgpu.ids" "$OUT/pytest.log" | tail -n 40
