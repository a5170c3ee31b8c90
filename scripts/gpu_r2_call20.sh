#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bench_path.py -m gpu -q --timeout 600 -p no:cacheprovider -k "rccl" > gpurun_out/pytest_sel.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_sel.log
grep -v "Warning\|warn" gpurun_out/pytest_sel.log | tail -n 50
# world-size-1 torchrun of the bench through the DP wrapper path is not possible (bench wraps only for world > 1); timing of the hooks:
timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", FLOWTRON_MFMA="bf16")
import torch, bench, flowtron, distributed as D
from flowtron_amd.optim import RAdam
D.init_distributed(0, 1, "nccl", None)
m = flowtron.Flowtron(**bench.MODEL_CONFIG); bench.init_weights(m, 1); m = m.cuda().train()
opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
bc = bench.synth_batch(32, 1241); T, Lk = bc["mel"].shape[2], bc["text"].shape[1]
b = {k: v.cuda() for k, v in bc.items()}; prior = bench.beta_binomial_prior_batch(bc["in_lens"], bc["out_lens"], T, Lk).cuda()
def step():
    opt.zero_grad(); out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], prior)
    nll, gl, ctc = crit(out, b["gate"], b["in_lens"], b["out_lens"]); (nll + gl + 0.01 * ctc).backward(); opt.clip_grad_norm_(1.0); opt.step()
def timeit(n=4):
    step(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("plain      ms/step", round(timeit(), 2))
m = D.apply_gradient_allreduce(m)
print("dp(ws=1, per-flow buckets, RCCL AVG) ms/step", round(timeit(), 2), m._grad_bucket_log)
PY
