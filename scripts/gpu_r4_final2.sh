#!/bin/bash
# Round-4 closing run: full -m gpu suite, smoke(), default bench line, LibriTTS lines (bf16 / fp16) -- the final code.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_final2"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=6 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
timeout 500 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
timeout 300 python bench.py --config libritts --steps 40 --warmup 5 > "$OUT/bench_line_libritts.json" 2> "$OUT/bench_libritts.err"
timeout 300 python bench.py --config libritts_fp16 --steps 40 --warmup 5 > "$OUT/bench_line_libritts_fp16.json" 2> "$OUT/bench_libritts_fp16.err"
tail -n 12 "$OUT/pytest_gpu.log"; tail -n 2 "$OUT/smoke.log"
python -c "
import json
for n in ('bench_line','bench_line_libritts','bench_line_libritts_fp16'):
    d=json.load(open('$OUT/%s.json' % n)); print(n, d['ms_per_step'], d['value'], d.get('skipped_steps'))"
