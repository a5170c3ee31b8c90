#!/bin/bash
# Round-3 closing confirmation (after the ABI v6 / oracle / test additions): full -m gpu suite, smoke(), default bench line.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3_final2"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 --durations=12 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
tail -n 22 "$OUT/pytest_gpu.log"
tail -n 3 "$OUT/smoke.log"
python -c "import json; d=json.load(open('$OUT/bench_line.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline'], d.get('parity'))"
tail -n 3 "$OUT/bench.err"
