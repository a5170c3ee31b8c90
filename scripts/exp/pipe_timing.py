"""Host-enqueue vs device time of one training step under different pipeline settings (bench workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ["FLOWTRON_MFMA"] = "bf16"
import bench, flowtron
from flowtron_amd.optim import RAdam
torch.manual_seed(1234)
model = flowtron.Flowtron(**bench.MODEL_CONFIG); bench.init_weights(model, 1234); model = model.cuda().train()
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
opt = RAdam(model.parameters(), lr=1e-3, weight_decay=1e-6)
bc = bench.synth_batch(32, 1241)
T, Lk = bc["mel"].shape[2], bc["text"].shape[1]
prior = bench.beta_binomial_prior_batch(bc["in_lens"], bc["out_lens"], T, Lk).cuda()
b = {k: v.cuda() for k, v in bc.items()}
def step(timing=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    out = model(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], prior)
    nll, gl, ctc = crit(out, b["gate"], b["in_lens"], b["out_lens"])
    loss = nll + gl + 0.01 * ctc
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    opt.clip_grad_norm_(1.0); opt.step(); torch.cuda.synchronize(); t5 = time.perf_counter()
    if timing is not None:
        timing.append([(t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3, (t5 - t4) * 1e3])
for pipe, ch, graph in (("0", "96", "1"), ("1", "96", "1"), ("1", "96", "0"), ("1", "216", "1"), ("1", "48", "1"), ("1", "431", "1")):
    os.environ.update(FLOWTRON_PIPELINE=pipe, FLOWTRON_CHUNK=ch, FLOWTRON_LSTM_GRAPH=graph)
    for _ in range(2): step()
    tm = []
    for _ in range(3): step(tm)
    a = [sum(x[i] for x in tm) / len(tm) for i in range(5)]
    print("pipe=%s chunk=%s graph=%s | fwd host %.1f total %.1f | bwd host %.1f total %.1f | opt %.1f | step(sync'd phases) %.1f ms" % (pipe, ch, graph, a[0], a[1], a[2], a[3], a[4], a[1] + a[3] + a[4]), flush=True)
