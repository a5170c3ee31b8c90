#!/bin/bash
# PMC look at the two image GEMM kernels on the gx0 forward shape: LDS bank conflicts, wait classes, MFMA busy.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$REPO/gpurun_out/pmc_gemm"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  TAG=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_run
  GEMM_BENCH_SHAPES=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- python "$REPO/scripts/exp/gemm_img_bench.py" > "$REPO/gpurun_out/pmc_gemm/run_$TAG.log" 2>&1
  F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$REPO/gpurun_out/pmc_gemm/pmc_$TAG.json" | grep -i gemm
done
