#!/bin/bash
# Round 5: ljs_cumm step time against the side-stream configuration of the backward's weight-gradient GEMMs.  usage: gpu_r5_cumm_sweep.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-sweep}"
OUT="$REPO/gpurun_out/r5_$TAG"
mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
for cfg in "0 12" "1 4" "1 8" "1 16" "1 24"; do
    set -- $cfg
    FT_CUMM_OVERLAP=$1 FT_CUMM_SIDE_CUS=$2 timeout 300 python bench.py --config ljs_cumm --steps 3 --warmup 1 --no-infer --no-trainpy --no-cpu-baseline 2>&1 >/dev/null | grep "timed region" | sed "s/^/overlap=$1 side_cus=$2: /" | tee -a "$OUT/sweep.log"
done
