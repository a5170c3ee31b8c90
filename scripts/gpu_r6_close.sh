#!/bin/bash
# round 6, closing run of the final code on one box: the whole GPU suite (figures printed: -rP), the PMC passes and the kernel-trace
# profile of a step, smoke, then the default bench line -- which reads the PMC summaries and the pytest log of THIS run (copied into
# the snapshot's profiles/ first; the caller copies the same files from gpurun_out/ into the tracked profiles/).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-final}"
OUT="$REPO/gpurun_out/r6_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rP > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -v "amdgpu.ids" "$OUT/pytest_gpu.log" | grep -E "^[0-9]+ passed|^FAILED|^ERROR| failed" | tail -n 30
rm -f profiles/*pytest_gpu*.log; cp "$OUT/pytest_gpu.log" "profiles/r06_${TAG}_pytest_gpu.log"
bash scripts/gpu_r6_pmc.sh > "$OUT/pmc.log" 2>&1; tail -n 3 "$OUT/pmc.log"
cp gpurun_out/pmc_r6/r06_pmc_*.json profiles/ 2>/dev/null
bash scripts/profile_r6.sh "$TAG" > "$OUT/profile.log" 2>&1; head -n 12 "$OUT/profile.log"
cd "$REPO"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -n 2 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"; tail -c 2500 "$OUT/bench_line.json"
for cfg in libritts libritts_fp16; do timeout 600 python bench.py --config $cfg --steps 30 --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_$cfg.json" 2>> "$OUT/bench_stderr.log"; python -c "import json,sys; d=json.loads(open('$OUT/bench_line_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"; done
timeout 900 python bench.py --config ljs_cumm --no-cpu-baseline --no-infer --no-trainpy > "$OUT/bench_line_ljs_cumm.json" 2>> "$OUT/bench_stderr.log"; python -c "import json; d=json.loads(open('$OUT/bench_line_ljs_cumm.json').read().strip().splitlines()[-1]); print('ljs_cumm', d['value'], d['ms_per_step'])"
