#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_l"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_optim.py -m gpu -x -q -k "gate_layer or persistent or failed_persistent or bilstm" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
tail -n 6 "$OUT/pytest.log"
timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -n 2 "$OUT/bench.err"
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'])"
cd /tmp
rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o bench -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/train_under_rocprof.log" 2>&1
TR=$(find /tmp/kt2 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/step_timeline.py "$TR" 1 > "$OUT/step_timeline.txt" 2>&1
head -n 4 "$OUT/step_timeline.txt"
grep -n "gemv" "$OUT/step_timeline.txt" | head -4
