#!/bin/bash
# Round 5: fused cumulative attention -- op-level parity + model-level tests, then the ljs_cumm line under rocprofv3.  usage: gpu_r5_cumm2.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-cumm2}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q --timeout 300 -p no:cacheprovider -k "fused_cumulative or cumm or cumulative" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -n 6 "$OUT/pytest.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o cumm -- \
    python "$REPO/bench.py" --config ljs_cumm --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_ljs_cumm_under_rocprof.json" 2> "$OUT/bench.err"
echo "rocprof exit $?" >> "$OUT/bench.err"
cd "$OUT"
find . -name "*kernel_trace.csv" -exec rm {} \;
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -n 12 "$f"; done
tail -n 4 bench.err
