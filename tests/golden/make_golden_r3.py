"""Round-3 golden vectors from the REAL reference (/root/reference) on CPU.

    python tests/golden/make_golden_r3.py                 # -> tests/golden/cfg1_t862_bf16.pt  (~10-20 min on 8 cores)
    python tests/golden/make_golden_r3.py --cumm          # -> tests/golden/cumm_full.pt        (~5 min)

cfg1_t862_bf16.pt  BASELINE configs[1] at the length bench.py times: the first four utterances (sorted by text length, incl. the
                 T = 862 one) of bench.synth_batch(32, 1234 + 7), 2-flow LJS config.json model (H = 1024), weights
                 synth.make_state_dict(seed 31), prior + CTC on -- the reference in fp32 AND under torch.autocast("cpu", bfloat16).
                 Per parameter: fp32 gradient norm, a seeded sample of the fp32 gradient, and the relative L2 deviation of the
                 reference's OWN bf16-autocast gradient from its fp32 gradient AT THIS SEQUENCE LENGTH (cfg2_bf16.pt has T <= 120;
                 rounding errors accumulate over 862 recurrent steps for everybody) -- the yardstick of
                 tests/test_gpu_bench_path.py::test_bf16_benchmark_config_at_its_own_shape_vs_oracle.
cumm_full.pt     use_cumm_attention = True (location-sensitive attention, flowtron.py:129-152, 697-723, 793-806) at FULL width
                 (H 1024, A 640, E 640), B = 2, T = 400 / 333, L = 31 / 24: strided forward outputs, losses, gradient norms and
                 samples, 48-frame infer mel -- pins row a17 of SURVEY section 8 beyond the H = 64 toy of small_cumm.pt.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import refshim, synth  # noqa: E402
from make_golden_r2 import sample_idx_stable  # noqa: E402

T862_SEED, T862_N = 31, 4


def t862_case():
    import bench
    bb = bench.synth_batch(32, 1234 + 7)
    n = T862_N
    out_lens, in_lens = bb["out_lens"][:n], bb["in_lens"][:n]
    T, Lk = int(out_lens.max()), int(in_lens.max())
    prior = bench.beta_binomial_prior_batch(in_lens, out_lens, T, Lk)
    batch = dict(mel=bb["mel"][:n, :, :T].contiguous(), speaker_ids=bb["speaker_ids"][:n], text=bb["text"][:n, :Lk].contiguous(),
                 in_lens=in_lens, out_lens=out_lens, gate_target=bb["gate"][:n, :T].contiguous(), attn_prior=prior)
    return dict(bench.MODEL_CONFIG), synth.make_state_dict(dict(bench.MODEL_CONFIG), seed=T862_SEED), batch


def run_ref(R, cfg, sd, b, autocast):
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)
    m.train()
    crit = R.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    real = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = m(b["mel"].clone(), b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"],
                    None if b.get("attn_prior") is None else b["attn_prior"].clone())
            snap = [x.detach().float().clone() for x in out[4]]
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        (nll + gl + 0.01 * ctc).backward()
    finally:
        F.dropout = real
    return (nll.detach().float(), gl.detach().float(), ctc.detach().float()), \
        {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}, out, snap, m


def make_t862(R):
    cfg, sd, b = t862_case()
    t0 = time.time()
    l32, g32, _, _, _ = run_ref(R, cfg, sd, b, False)
    print("fp32 reference: %.0f s, losses" % (time.time() - t0), [x.item() for x in l32], flush=True)
    t0 = time.time()
    l16, g16, _, _, _ = run_ref(R, cfg, sd, b, True)
    print("bf16-autocast reference: %.0f s, losses" % (time.time() - t0), [x.item() for x in l16], flush=True)
    res = {"seed": T862_SEED, "n_utt": T862_N, "out_lens": b["out_lens"].tolist(), "in_lens": b["in_lens"].tolist(),
           "losses_fp32": l32, "losses_bf16_autocast": l16, "grad": {}}
    for k, g in g32.items():
        flat = g.reshape(-1)
        idx = sample_idx_stable(flat.numel(), k)
        dev = (g16[k] - g).norm().item() / max(g.norm().item(), 1e-30)
        res["grad"][k] = {"norm": flat.norm().item(), "idx": idx, "sample": (flat if idx is None else flat[idx]).clone(),
                          "ref_bf16_autocast_rel_dev": dev}
    out = os.path.join(HERE, "cfg1_t862_bf16.pt")
    torch.save(res, out)
    print(out, os.path.getsize(out) // 1024, "KiB")
    for d, k in sorted(((e["ref_bf16_autocast_rel_dev"], k) for k, e in res["grad"].items()), reverse=True)[:24]:
        print("   %.4f %s" % (d, k))


CUMM_LENS = ([400, 333], [31, 24])
CUMM_SEED = 9


def make_cumm(R):
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, use_cumm_attention=True)
    sd = synth.make_state_dict(cfg, seed=CUMM_SEED)
    b = synth.make_batch(cfg, CUMM_LENS[0], CUMM_LENS[1], seed=CUMM_SEED, with_prior=True)
    t0 = time.time()
    losses, g, out, snap_lp, m = run_ref(R, cfg, sd, b, False)
    print("cumulative-attention reference (fwd + bwd): %.0f s, losses" % (time.time() - t0), [x.item() for x in losses], flush=True)
    st = 8
    z, log_s, gate, attn = out[0], out[1], out[2], out[3]
    res = {"cfg": cfg, "seed": CUMM_SEED, "out_lens": CUMM_LENS[0], "in_lens": CUMM_LENS[1], "stride": st,
           "nll": losses[0], "gate_loss": losses[1], "ctc": losses[2],
           "z": z.detach()[::st].clone(), "log_s": [x.detach()[::st].clone() for x in log_s], "gate": gate.detach()[::st].clone(),
           "attn": [x.detach()[:, ::st].clone() for x in attn], "logprob": [x[:, ::st].clone() for x in snap_lp],
           "grad_norm": {k: v.norm().item() for k, v in g.items()},
           "grad_sample": {k: v.flatten()[:: max(1, v.numel() // 64)][:64].clone() for k, v in g.items()}}
    m.eval()
    n_infer = 48
    rs = np.random.RandomState(CUMM_SEED + 11)
    residual = torch.from_numpy(rs.standard_normal((1, cfg["n_mel_channels"], n_infer)).astype(np.float32)) * 0.5
    with torch.no_grad():
        mel, att = m.infer(residual.clone(), b["speaker_ids"][:1], b["text"][:1, : CUMM_LENS[1][0]], gate_threshold=1.0)
    res.update(infer_mel=mel, infer_attn=[torch.cat(a)[:, 0] for a in att])
    outp = os.path.join(HERE, "cumm_full.pt")
    torch.save(res, outp)
    print(outp, os.path.getsize(outp) // 1024, "KiB")


def main():
    assert refshim.available(), "needs /root/reference"
    torch.set_num_threads(8)
    R = refshim.load()
    if "--cumm" in sys.argv:
        make_cumm(R)
    else:
        make_t862(R)


if __name__ == "__main__":
    main()
