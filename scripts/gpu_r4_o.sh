#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_o"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for E in 0 1; do
FT_MS_EXP=$E FWD_NGS=31 BWD_NGS=21 PROF_NG=31 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench_$E.log" 2>&1
echo "EXP $E"; grep "ng=31\|wave 0 (ng 31" "$OUT/persist_bench_$E.log"
done
