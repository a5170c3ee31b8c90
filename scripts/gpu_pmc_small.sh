#!/bin/bash
# PMC passes (HBM traffic) on the stand-alone LSTM step launches only (scripts/exp/lstm_only.py, ~170 dispatches).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$REPO/gpurun_out/pmc"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python "$REPO/scripts/exp/lstm_only.py" > "$REPO/gpurun_out/pmc/run_$C.log" 2>&1
  echo "rocprof pmc $C exit $?" >> "$REPO/gpurun_out/pmc/run_$C.log"
  F=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && head -n 3 "$F" > "$REPO/gpurun_out/pmc/head_$C.csv" && python "$REPO/scripts/pmc_summarize.py" "$F" "$REPO/gpurun_out/pmc/summary_$C.json"
done
