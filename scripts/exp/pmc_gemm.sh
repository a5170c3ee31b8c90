#!/bin/bash
# PMC passes over the image-GEMM bench (scripts/exp/gemm_img_bench.py, first shape): where do the waves of the 256^2 kernels wait?
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/gemm"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_run
  GEMM_BENCH_SHAPES=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_run -o pmc -- python $REPO/scripts/exp/gemm_img_bench.py > "$OUT/pmc_gemm_$i.log" 2>&1
  echo "rocprof pmc [$C] exit $?" >> "$OUT/pmc_gemm_$i.log"
  F=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python "$REPO/scripts/pmc_summarize.py" "$F" "$OUT/pmc_gemm_$i.json" > /dev/null
  python - "$OUT/pmc_gemm_$i.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'gemm_bf16' in k: print(k[:80].replace('void (anonymous namespace)::',''), {c:round(x['avg']) for c,x in v.items()})
PY
done
