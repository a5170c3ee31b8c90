"""B = 40 > 32 through the whole model: sliced persistent launches (FLOWTRON_LSTM_PERSIST_WIDE, ops.lstm_persist_slices) against the
launch-per-step / two-layer wavefront kernels on the same weights and batch: z bit-identical (the persistent forward is), gradients to
the reduce-scatter backward's rounding.  usage: python scripts/exp/wide_batch_check.py"""
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
B = 40
rs = np.random.RandomState(1)
out_lens = sorted((int(x) for x in rs.randint(60, 200, size=B)), reverse=True)
in_lens = sorted((max(5, t // 6) for t in out_lens), reverse=True)
batch = synth.make_batch(cfg, out_lens, in_lens, seed=10, with_prior=True)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
m = flowtron.Flowtron(**cfg); m.load_state_dict(sd); m = m.to(dev).eval()

def run(wide):
    ops._PERSIST_WIDE = wide
    for p in m.parameters(): p.grad = None
    n0 = ops.PERSIST_LAUNCHES
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    torch.cuda.synchronize()
    assert ops.check_persist_status()
    return out[0].detach().clone(), {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}, ops.PERSIST_LAUNCHES - n0, float(nll)

z1, g1, n1, l1 = run(True)
z0, g0, n0, l0 = run(False)
worst = sorted(((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0 if "conv.bias" not in k), reverse=True)[:4]
print("B = %d: persistent launches wide %d / off %d; nll %.6f / %.6f; z bit-identical %s (max diff %.2e); worst gradient rel-L2 %s" %
      (B, n1, n0, l1, l0, torch.equal(z1, z0), float((z1 - z0).abs().max()), ["%s %.1e" % (k[-40:], w) for w, k in worst]))
