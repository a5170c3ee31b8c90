#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_l"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-infer > "$OUT/bench_under_rocprof.log" 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
head -n 2 "$F" > "$OUT/trace_head.csv"
python $REPO/scripts/kernel_trace_table.py "$F" 100 > "$OUT/trace_table.txt" 2>&1
grep -v "lstm_fwd_step\|lstm_bwd_step" "$OUT/trace_table.txt" | head -n 60
