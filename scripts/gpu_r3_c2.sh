#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -k "cumm or cumulative" -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_c2.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_c2.log
timeout 300 python bench.py --config ljs_cumm --steps 3 --warmup 1 --no-infer --no-cpu-baseline > gpurun_out/bench_c2_cumm.json 2> gpurun_out/bench_c2_cumm.err
echo "bench cumm exit $?" >> gpurun_out/bench_c2_cumm.err
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o bench -- python $REPO/bench.py --config ljs_cumm --steps 1 --warmup 0 --no-cpu-baseline --no-infer > $REPO/gpurun_out/bench_c2_rocprof.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/bench_c2_cumm_kernel_stats.csv 2>/dev/null
cd "$REPO"
grep -E "passed|failed|cumulative attention|FAILED|Error|error" gpurun_out/pytest_c2.log | head -20
tail -n 4 gpurun_out/bench_c2_cumm.err
head -c 400 gpurun_out/bench_c2_cumm.json; echo
head -n 22 gpurun_out/bench_c2_cumm_kernel_stats.csv | cut -c1-170
