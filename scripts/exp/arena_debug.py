"""debug: gradients of the full-width model on a tiny batch with and without the flat arena (weight_grad_out)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FLOWTRON_MFMA", "bf16")
import numpy as np, torch
import flowtron
from flowtron_amd.optim import RAdam
from flowtron_amd import ops
from oracle import synth

dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_MODEL_CONFIG); cfg["n_flows"] = 2
sd = synth.make_state_dict(cfg, seed=3)
batch = synth.make_batch(cfg, [40, 33, 21], [12, 9, 7], seed=10, with_prior=True)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)

def grads(m):
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}

m0 = flowtron.Flowtron(**cfg); m0.load_state_dict(sd); m0 = m0.to(dev).eval()
local = grads(m0)
local2 = None
for p in m0.parameters(): p.grad = None
local2 = grads(m0)
m = flowtron.Flowtron(**cfg); m.load_state_dict(sd); m = m.to(dev).eval()
opt = RAdam(m.parameters(), lr=1e-3)
for tag, flag in (("arena", True), ("arena-off", False), ("arena", True)):
    ops._ARENA_GRADS = flag
    opt.zero_grad()
    g = grads(m)
    a = opt.arena
    inside = sum(1 for p in m.parameters() if a._ptr_lo <= p.grad.data_ptr() < a._ptr_hi)
    worst = sorted(((float(np.abs(g[k] - local[k]).max() / (np.abs(local[k]).max() + 1e-12)), k) for k in g), reverse=True)[:6]
    print(tag, "grads inside arena before adoption:", inside, "/", len(g), "pass_clean", a._pass_clean, "handed", len(a._handed))
    for w, k in worst: print("   %-50s %.3e" % (k, w))
worst = sorted(((float(np.abs(local2[k] - local[k]).max() / (np.abs(local[k]).max() + 1e-12)), k) for k in local), reverse=True)[:4]
print("run-to-run (no arena):", worst)
