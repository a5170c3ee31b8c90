#!/bin/bash
# Round-4 call B: the new guards / trajectory / DP tests and a bench line with the trainpy_step block.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_b"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_train_loop.py -m gpu -x -q --timeout 500 -p no:cacheprovider -s \
   -k "image_only or three_flows or trajectory" > "$OUT/pytest_new.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_new.log"
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q --timeout 500 -p no:cacheprovider > "$OUT/pytest_dist.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_dist.log"
timeout 400 python bench.py --steps 20 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
tail -n 12 "$OUT/pytest_new.log"; tail -n 6 "$OUT/pytest_dist.log"; tail -n 5 "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d.get('trainpy_step'), indent=1)); print(d['roofline']['dominant_kernel']['us_per_step'], d['roofline']['second_kernel']['us_per_step'], d['roofline']['gemm_mfma_busy_frac_pmc'])"
