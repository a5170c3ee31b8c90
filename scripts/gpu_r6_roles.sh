#!/bin/bash
# round 6: probe of the R / window / role persistent recurrences (scripts/exp/lstm_roles_bench.py)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r6_${1:-roles}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python scripts/exp/lstm_roles_bench.py > "$OUT/lstm_roles_bench.log" 2>&1
echo "exit $?" >> "$OUT/lstm_roles_bench.log"
cp gpurun_out/lstm_roles_bench.json "$OUT/" 2>/dev/null
grep -v "amdgpu.ids" "$OUT/lstm_roles_bench.log" | tail -n 45
