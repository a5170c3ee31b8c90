#!/bin/bash
# Round 6: kernel-trace stats + ordered step timeline of the training step on the final code (the PMC passes of
# profiles/r05_pmc_* stand: the GEMM / recurrence / attention kernels are unchanged).   usage: profile_r5b.sh [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_r6${1:+_$1}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o bench -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/train_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/train_under_rocprof.log"
TR=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) "$OUT/train_kernel_stats.csv" 2>/dev/null
grep '^{' "$OUT/train_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
for k in 1 2 3 4; do python $REPO/scripts/step_timeline.py "$TR" $k > "$OUT/step_timeline_$k.txt" 2>&1; head -n 1 "$OUT/step_timeline_$k.txt"; done
BEST=$(python -c "
import re
best=None
for k in (1,2,3,4):
    m=re.search(r'span ([0-9.]+) ms', open('$OUT/step_timeline_%d.txt' % k).readline())
    if m and (best is None or float(m.group(1)) < best[0]): best=(float(m.group(1)), k)
print(best[1] if best else 1)")
cp "$OUT/step_timeline_$BEST.txt" "$OUT/step_timeline.txt"
python $REPO/scripts/kernel_trace_table.py "$TR" 30 > "$OUT/kernel_instances.txt" 2>&1
head -n 45 "$OUT/step_timeline.txt"
