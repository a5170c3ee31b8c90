#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_m"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 120 python scripts/prof_stft.py > "$OUT/prof_stft.log" 2>&1; tail -n 2 "$OUT/prof_stft.log"
timeout 900 python -m pytest tests -m gpu -x -q -k "stft or mel or audio or gate_layer" -p no:cacheprovider > "$OUT/pytest_a.log" 2>&1
tail -n 4 "$OUT/pytest_a.log"
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_loop.py tests/test_gpu_bench_path.py -m gpu -x -q -p no:cacheprovider > "$OUT/pytest_b.log" 2>&1
tail -n 6 "$OUT/pytest_b.log"
cd /tmp
rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/train_under_rocprof.log" 2>&1
TR=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
head -n 2 "$OUT/step_timeline.txt"
grep -n "gemv" "$OUT/step_timeline.txt" | head -2
