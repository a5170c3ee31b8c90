#!/bin/bash
# weight images of a forward in one launch (FLOWTRON_WEIGHT_TABLE): A/B on the step, then the full suite + default line
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r5b_${1:-j}"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for m in 1 0 1 0; do
    echo "WEIGHT_TABLE=$m: $(FLOWTRON_WEIGHT_TABLE=$m timeout 300 python bench.py --steps 40 --warmup 3 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/weight_table_ab.log"
done
echo "B=64: $(timeout 300 python bench.py --batch 64 --steps 10 --warmup 2 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep 'timed region')" | tee -a "$OUT/weight_table_ab.log"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "passed|failed|FAILED|ERROR|Error" | tail -n 6
