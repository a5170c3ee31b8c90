#!/bin/bash
# Round-3 run C (GPU box, via gpurun): the cumulative-attention sequence op (goldens, library walk vs python walk, bench line)
# and the default bench line with the backward recurrence on the BARE hand-off.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -k "cumm or cumulative" tests/test_gpu_ops.py::test_persistent_lstm_backward_is_bit_identical_to_launch_per_step -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_c.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_c.log
timeout 400 python bench.py --config ljs_cumm --steps 3 --warmup 1 --no-infer --no-cpu-baseline > gpurun_out/bench_c_cumm.json 2> gpurun_out/bench_c_cumm.err
echo "bench cumm exit $?" >> gpurun_out/bench_c_cumm.err
timeout 400 python bench.py --no-infer > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
echo "bench exit $?" >> gpurun_out/bench_c.err
grep -E "passed|failed|cumulative attention|FAILED|Error|error" gpurun_out/pytest_c.log | head -40
tail -n 4 gpurun_out/bench_c_cumm.err gpurun_out/bench_c.err
head -c 700 gpurun_out/bench_c_cumm.json; echo
head -c 1500 gpurun_out/bench_c.json; echo
