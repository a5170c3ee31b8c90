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
def step():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    out = model(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], prior)
    nll, gl, ctc = crit(out, b["gate"], b["in_lens"], b["out_lens"])
    loss = nll + gl + 0.01 * ctc
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    opt.clip_grad_norm_(1.0); opt.step(); torch.cuda.synchronize()
    ms = torch.cuda.memory_stats()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3, ms.get("num_device_alloc", -1), ms.get("num_sync_all_streams", -1)
os.environ.update(FLOWTRON_PIPELINE="1", FLOWTRON_CHUNK=sys.argv[1] if len(sys.argv) > 1 else "216", FLOWTRON_LSTM_GRAPH="1")
for it in range(10):
    r = step()
    print("it %d: fwd host %.1f total %.1f | bwd host %.1f total %.1f | device_allocs %d sync_all %d" % ((it,) + r), flush=True)
