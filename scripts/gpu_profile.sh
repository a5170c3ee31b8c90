#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (GPU box, via gpurun). Summaries land in gpurun_out/prof/.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$REPO/gpurun_out/prof"
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -o bench -- \
    python "$REPO/bench.py" --steps ${PROF_STEPS:-2} --warmup 1 --no-cpu-baseline --no-infer > "$REPO/gpurun_out/prof/bench_under_rocprof.log" 2>&1
echo "rocprof exit $?" >> "$REPO/gpurun_out/prof/bench_under_rocprof.log"
cd "$REPO/gpurun_out/prof"
find . -name "*kernel_trace.csv" -size +20M -exec sh -c 'head -n 2000 "$1" > "$1.head"; rm "$1"' _ {} \;
find . -type f | head -50
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -n 25 "$f"; done
tail -n 3 bench_under_rocprof.log
