#!/bin/bash
# Round 5: fused cumulative attention -- parity tests, stage stamps, the ljs_cumm bench line.  usage: gpu_r5_cumm3.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-cumm3}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q --timeout 300 -p no:cacheprovider -k "fused_cumulative or cumm or cumulative" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
tail -n 6 "$OUT/pytest.log"
timeout 300 python scripts/exp/cumm_prof.py 60 > "$OUT/cumm_stage_stamps.log" 2>&1
tail -n 5 "$OUT/cumm_stage_stamps.log"
timeout 600 python bench.py --config ljs_cumm --steps 3 --warmup 1 --no-infer --no-trainpy --no-cpu-baseline > "$OUT/bench_line_ljs_cumm.json" 2> "$OUT/bench_cumm.err"
echo "bench exit $?" >> "$OUT/bench_cumm.err"
tail -n 4 "$OUT/bench_cumm.err"
