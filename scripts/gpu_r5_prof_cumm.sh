#!/bin/bash
# Round 5: rocprofv3 kernel stats of the ljs_cumm bench line.  usage: gpu_r5_prof_cumm.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-cummprof}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o cumm -- \
    python "$REPO/bench.py" --config ljs_cumm --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof exit $?" >> "$OUT/bench_under_rocprof.log"
cd "$OUT"
find . -name "*kernel_trace.csv" -size +20M -exec sh -c 'head -n 4000 "$1" > "$1.head"; rm "$1"' _ {} \;
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -n 22 "$f"; done
tail -n 3 bench_under_rocprof.log
