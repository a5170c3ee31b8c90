#!/bin/bash
# Round 6: step timeline of the train.py call sequence (bench.py's trainpy_step leg, which runs after the headline loop): where does
# the GPU wait for the host?   usage: profile_trainpy.sh [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_trainpy${1:+_$1}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt3 && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -o bench -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-infer > "$OUT/trainpy_under_rocprof.log" 2>&1
echo "rocprof kernel-trace exit $?" >> "$OUT/trainpy_under_rocprof.log"
TR=$(find /tmp/kt3 -name "*kernel_trace.csv" | head -1)
grep '^{' "$OUT/trainpy_under_rocprof.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
for k in 1 2; do python $REPO/scripts/step_timeline.py "$TR" $k > "$OUT/trainpy_step_timeline_$k.txt" 2>&1; head -n 2 "$OUT/trainpy_step_timeline_$k.txt"; done
sed -n '/gaps >= 20 us/,/--- timeline/p' "$OUT/trainpy_step_timeline_1.txt" | head -60
