import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import flowtron
from oracle import synth
from flowtron_amd import ops
from test_gpu_model import build, cuda_batch
cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60)
b = cuda_batch(synth.make_batch(cfg, [48, 41, 17, 48], [14, 12, 12, 5], seed=6, with_prior=True))
res = {}
for name, mode, img, l2, bi in (("f32", "f32", True, "1", "1"), ("bf16_img", "bf16", True, "1", "1"), ("bf16_noimg", "bf16", False, "1", "1"),
                                ("bf16_old", "bf16", False, "0", "0")):
    ops._BF16_IMAGES = img
    os.environ["FLOWTRON_LSTM2"] = l2; os.environ["FLOWTRON_BILSTM"] = bi
    m, _ = build(cfg, 6, mode)
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).sum().backward()
    torch.cuda.synchronize()
    res[name] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    print(name, nll.item(), gl.item(), ctc.item())
for k, r in res["f32"].items():
    n = max(r.norm().item(), 1e-4 * r.numel() ** 0.5)
    print(f"{k:55s} |g|={r.norm().item():9.3e} " + " ".join(f"{v}={(res[v][k]-r).norm().item()/n:8.2e}" for v in ("bf16_img", "bf16_noimg", "bf16_old")))
