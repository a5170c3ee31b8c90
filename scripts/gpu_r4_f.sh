#!/bin/bash
# Round-4 call F: new GEMM / RS tests + step timeline of the current code
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r4_f"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bench_path.py -m gpu -x -q --timeout 500 -p no:cacheprovider -k "concatenated or two_inputs or ragged_batch_mel or bf16_benchmark_config or bf16_mode_gradients or edge_shapes" > "$OUT/pytest_sel.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_sel.log"
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_under_rocprof.log" 2>&1
TR=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
cd "$REPO"
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_train_only.json" 2> "$OUT/bench.err"
tail -n 5 "$OUT/pytest_sel.log"
head -n 45 "$OUT/step_timeline.txt"
python -c "import json; d=json.load(open('$OUT/bench_line_train_only.json')); print(d['ms_per_step'], d['value'])"
