#!/bin/bash
# PMC passes (HBM traffic counters) for the bench command; one counter group per pass (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md).  No trace domains besides kernel dispatch.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$REPO/gpurun_out/pmc"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- \
      python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-infer > "$REPO/gpurun_out/pmc/run_$C.log" 2>&1
  echo "rocprof pmc $C exit $?" >> "$REPO/gpurun_out/pmc/run_$C.log"
  F=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  ls -la /tmp/pmc_$C $(dirname "$F") >> "$REPO/gpurun_out/pmc/run_$C.log" 2>&1
  head -n 3 "$F" > "$REPO/gpurun_out/pmc/head_$C.csv"
  python "$REPO/scripts/pmc_summarize.py" "$F" "$REPO/gpurun_out/pmc/summary_$C.json"
done
