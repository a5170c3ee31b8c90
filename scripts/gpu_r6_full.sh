#!/bin/bash
# round 6: the whole GPU suite + smoke + the default bench line
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r6_${1:-full}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR" | tail -n 30
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -n 2 "$OUT/smoke.log"
if [ -z "$NO_BENCH" ]; then timeout 900 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"; tail -c 1500 "$OUT/bench_line.json"; fi
