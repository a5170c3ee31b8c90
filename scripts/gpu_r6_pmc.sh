#!/bin/bash
# round 6: PMC passes over one training step (one counter group per rocprofv3 run, kernel dispatch tracing only): HBM bytes, MFMA busy,
# and the VALU / transcendental issue counters the attention / CTC kernels are priced with (VERDICT r5 #3v)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_r6"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  TAG=$(echo $C | cut -d' ' -f1)
  [ "$TAG" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && TAG=MFMA_BUSY
  [ "$TAG" = "SQ_INSTS_VALU" ] && TAG=VALU_TRANS
  [ "$TAG" = "SQ_WAVE_CYCLES" ] && TAG=WAVE_CYCLES
  rm -rf /tmp/pmc_run
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-trainpy --no-infer > "$OUT/pmc_${TAG}.log" 2>&1
  echo "rocprof pmc [$C] exit $?" >> "$OUT/pmc_${TAG}.log"
  F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/r06_pmc_${TAG}.json" | head -3
done
ls -la "$OUT"
