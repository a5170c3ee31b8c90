import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ["FLOWTRON_MFMA"] = "bf16"
import flowtron
from oracle import synth
cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60)
b = synth.make_batch(cfg, [70, 61, 33, 70, 9], [14, 12, 12, 7, 3], seed=4, with_prior=True)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
def d(a, c): return (a - c).abs().max().item()
crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
res = {}
for pipe in ("0", "1", "1b"):
    os.environ.update(FLOWTRON_PIPELINE=pipe[0], FLOWTRON_CHUNK="16", FLOWTRON_LSTM_GRAPH="1")
    m = flowtron.Flowtron(**cfg); m.load_state_dict(synth.make_state_dict(cfg, 4)); m = m.cuda().eval()
    outs = []
    for it in range(3):
        m.zero_grad()
        out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
        outs.append([out[0].detach().clone(), out[1][0].detach().clone(), out[1][1].detach().clone(), out[3][0].detach().clone(), out[3][1].detach().clone()])
        if it < 2:
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).sum().backward()
        torch.cuda.synchronize()
    res[pipe] = outs
    print(pipe, "it1 vs it0:", [d(a, c) for a, c in zip(outs[1], outs[0])], "it2 vs it0:", [d(a, c) for a, c in zip(outs[2], outs[0])])
print("pipe it0 vs seq it0 [z, log_s0, log_s1, attn0, attn1]:", [d(a, c) for a, c in zip(res["1"][0], res["0"][0])])
print("pipe it1 vs seq it1:", [d(a, c) for a, c in zip(res["1"][1], res["0"][1])])
print("pipe1b it0 vs pipe it0:", [d(a, c) for a, c in zip(res["1b"][0], res["1"][0])])
