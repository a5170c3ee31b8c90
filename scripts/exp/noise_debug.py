"""debug: run-to-run noise of the bf16 gradients of the full-width model on a tiny batch (same weights, same batch, two passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FLOWTRON_MFMA", "bf16")
import numpy as np, torch
import flowtron
from flowtron_amd import ops
from oracle import synth

dev = torch.device("cuda", 0)
cfg = dict(synth.DEFAULT_MODEL_CONFIG); cfg["n_flows"] = 2
sd = synth.make_state_dict(cfg, seed=3)
lens = ([40, 33, 21], [12, 9, 7]) if len(sys.argv) < 2 else ([400, 333, 210, 150], [70, 60, 40, 22])
batch = synth.make_batch(cfg, lens[0], lens[1], seed=10, with_prior=True)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
m0 = flowtron.Flowtron(**cfg); m0.load_state_dict(sd); m0 = m0.to(dev).eval()

def grads():
    for p in m0.parameters(): p.grad = None
    out = m0(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m0.named_parameters()}, (float(nll), float(gl), float(ctc)), out[0].detach().float().cpu().numpy()

def report(tag):
    g0, l0, z0 = grads(); g1, l1, z1 = grads()
    rl2 = sorted(((float(np.linalg.norm(g1[k] - g0[k]) / (np.linalg.norm(g0[k]) + 1e-30)), k) for k in g0), reverse=True)
    print("%-28s losses equal %s z maxdiff %.2e | worst rel-L2:" % (tag, l0 == l1, float(np.abs(z0 - z1).max())), ["%s %.1e" % (k.replace("ar_step.", "")[-38:], w) for w, k in rl2[:5]])

report("default")
ops._ENC_SPLITK = True; report("encoder fwd split-K with atomics")
ops._ENC_SPLITK = False; report("encoder fwd split-K off")
ops.FUSED_LOSS = False; report("+ fused loss off")
ops.FUSED_LOSS = True
os.environ["FLOWTRON_LSTM_PERSIST"] = "0"; report("+ persistent recurrences off")
