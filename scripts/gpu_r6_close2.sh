#!/bin/bash
# round 6: PMC passes again (summaries with the largest dispatch per kernel) + the default bench line reading them
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r6_${1:-final}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
bash scripts/gpu_r6_pmc.sh > "$OUT/pmc.log" 2>&1; tail -n 3 "$OUT/pmc.log"
cp gpurun_out/pmc_r6/r06_pmc_*.json profiles/ 2>/dev/null
cd "$REPO"
timeout 900 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"; tail -c 600 "$OUT/bench_line.json"
