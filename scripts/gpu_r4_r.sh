#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_r"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for S in 0 2 4 6; do
FT_FWD_SLEEP=$S FWD_NGS=1 BWD_NGS=21 PROF_NG=1 timeout 300 python scripts/exp/lstm_persist_bench.py > "$OUT/persist_bench_$S.log" 2>&1
echo "SLEEP $S"; grep "persistent ng=1 \|^wave [02]" "$OUT/persist_bench_$S.log"
done
