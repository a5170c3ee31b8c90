"""Generate golden vectors by running the REAL reference (/root/reference) on CPU.

Run once in the build container:   python tests/golden/make_golden.py
Outputs small .pt fixtures next to this file.  Inputs and weights are NOT
stored: they are rebuilt from seeds by oracle/synth.py (numpy RandomState).

Fixtures:
  small_f2.pt   2-flow small model: every forward output, 3 losses, all grads, infer mel
  small_f3.pt   3-flow small model, no prior (odd last flow, log(p+1e-8) branch)
  small_cumm.pt 2-flow small model with use_cumm_attention=True (location-sensitive attention, flowtron.py:129-152, 697-723)
  cfg1_full.pt  BASELINE config 1 (1-flow, n_text=148, B=2, T=800/650, L=148/120, fp32):
                losses, strided z / log_s / attn slices, grad norms, 48-frame infer mel
  stft_mel.pt   reference STFT/TacotronSTFT with a stub librosa (filterbank = oracle's
                Slaney restatement: parity UNPINNED for the filterbank constants)
  prior.pt      scipy.stats.betabinom prior (data.py:31-41) for (P,M)=(13,40),(148,800)[::50]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import refshim, synth  # noqa: E402
from oracle import flowtron_oracle as O  # noqa: E402


def run_model_case(R, cfg, out_lens, in_lens, with_prior, seed, full_dump, n_infer):
    sd = synth.make_state_dict(cfg, seed=seed)
    b = synth.make_batch(cfg, out_lens, in_lens, seed=seed, with_prior=with_prior)
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)
    crit = R.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    # gradient parity needs train() with dropout neutralised
    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    try:
        m.train()
        prior = None if not with_prior else b["attn_prior"].clone()
        out = m(b["mel"].clone(), b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], prior)
        # snapshot BEFORE the loss: FlowtronLoss rolls odd flows' attn_logprob in place
        # (flowtron.py:252-255) and never restores the caller's tensor
        snap_lp = [x.detach().clone() for x in out[4]]
        nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        total = nll + gl + 0.01 * ctc
        total.backward()
    finally:
        F.dropout = real_dropout
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    res = {"cfg": cfg, "seed": seed, "out_lens": list(out_lens), "in_lens": list(in_lens), "with_prior": with_prior,
           "nll": nll.detach(), "gate_loss": gl.detach(), "ctc": ctc.detach()}
    z, log_s, gate, attn, lp = [o for o in out[:5]]
    lp = snap_lp
    if full_dump:
        res.update(z=z.detach(), log_s=[x.detach() for x in log_s], gate=gate.detach(),
                   attn=[x.detach() for x in attn], logprob=[x.detach() for x in lp], grads=g)
    else:
        st = 8
        res.update(z=z.detach()[::st].clone(), log_s=[x.detach()[::st].clone() for x in log_s],
                   gate=gate.detach()[::st].clone(),
                   attn=[x.detach()[:, ::st].clone() for x in attn], logprob=[x.detach()[:, ::st].clone() for x in lp],
                   stride=st,
                   grad_norm={k: v.norm().item() for k, v in g.items()},
                   grad_sample={k: v.flatten()[:: max(1, v.numel() // 64)][:64].clone() for k, v in g.items()})
    # inference (eval, gate disabled and enabled)
    m.eval()
    rs = np.random.RandomState(seed + 11)
    residual = torch.from_numpy(rs.standard_normal((1, cfg["n_mel_channels"], n_infer)).astype(np.float32)) * 0.5
    txt = b["text"][:1, : in_lens[0]]
    spk = b["speaker_ids"][:1]
    with torch.no_grad():
        mel, att = m.infer(residual.clone(), spk, txt, gate_threshold=1.0)
        mel_g, _ = m.infer(residual.clone(), spk, txt, gate_threshold=0.5)
    res.update(infer_mel=mel, infer_attn=[torch.cat(a)[:, 0] for a in att], infer_gated_frames=mel_g.shape[2])
    if not cfg.get("use_cumm_attention", False):
        pr = O.beta_binomial_prior(in_lens[0], n_infer).float()[None]          # [1, N, L] prior at inference (flowtron.py:799)
        with torch.no_grad():
            mel_p, att_p = m.infer(residual.clone(), spk, txt, gate_threshold=1.0, attn_prior=pr)
        res.update(infer_prior_mel=mel_p, infer_prior_attn=[torch.cat(a)[:, 0] for a in att_p])
    return res


def stub_librosa():
    lib = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    filt = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        lengths = [(0, 0)] * data.ndim
        lengths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, lengths)

    util.pad_center = pad_center
    util.tiny = lambda x: np.finfo(np.float32).tiny
    util.normalize = lambda x, **kw: x
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax).numpy()
    lib.util, lib.filters = util, filt
    sys.modules.update({"librosa": lib, "librosa.util": util, "librosa.filters": filt})


def main():
    assert refshim.available(), "needs /root/reference"
    torch.set_num_threads(8)
    R = refshim.load()
    small = dict(synth.SMALL_MODEL_CONFIG)
    torch.save(run_model_case(R, small, [23, 17, 20], [9, 7, 5], True, 5, True, 14), os.path.join(HERE, "small_f2.pt"))
    torch.save(run_model_case(R, dict(small, n_flows=3), [15, 11], [6, 6], False, 6, True, 10), os.path.join(HERE, "small_f3.pt"))
    torch.save(run_model_case(R, dict(small, use_cumm_attention=True), [13, 9, 11], [7, 5, 4], True, 8, True, 9),
               os.path.join(HERE, "small_cumm.pt"))
    cfg1 = dict(synth.DEFAULT_MODEL_CONFIG, n_flows=1, n_text=148)
    torch.save(run_model_case(R, cfg1, [800, 650], [148, 120], True, 1234, False, 48), os.path.join(HERE, "cfg1_full.pt"))

    # audio front end
    stub_librosa()
    sys.path.insert(0, refshim.REF_DIR)
    import audio_processing as AP  # the reference module
    stft = AP.TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0)
    y = torch.stack([synth.make_audio(256 * 40, seed=s) for s in (0, 1)])
    mel = stft.mel_spectrogram(y)
    mag, _ = stft.stft_fn.transform(y)
    torch.save({"n_samples": 256 * 40, "seeds": [0, 1], "mel": mel, "mag_b0_f7": mag[0, :, 7].clone()},
               os.path.join(HERE, "stft_mel.pt"))

    # beta-binomial prior (data.py:31-41) straight from scipy like the reference
    from scipy.stats import betabinom

    def ref_prior(P, M, s=1.0):
        x = np.arange(0, P)
        return torch.tensor(np.array([betabinom(P - 1, s * i, s * (M + 1 - i)).pmf(x) for i in range(1, M + 1)]))

    torch.save({"p13_m40": ref_prior(13, 40), "p148_m800_s50": ref_prior(148, 800)[::50].clone()},
               os.path.join(HERE, "prior.pt"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
