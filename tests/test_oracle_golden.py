"""Pins the CPU oracle (oracle/flowtron_oracle.py) against golden vectors that
tests/golden/make_golden.py produced by executing the REAL reference
(/root/reference/flowtron.py, audio_processing.py, scipy betabinom)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import flowtron_oracle as O
from oracle import synth

TOL = 2e-5


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def _maxdiff(a, b):
    return (a - b).abs().max().item()


@pytest.mark.parametrize("name", ["small_f2.pt", "small_f3.pt", "small_cumm.pt", "small_dummy_spk.pt"])
def test_forward_loss_grads_small(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = g["cfg"]
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.make_state_dict(cfg, seed=g["seed"]).items()}
    b = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=g["with_prior"])
    out = O.forward(sd, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    assert _maxdiff(out[0], g["z"]) < TOL
    assert _maxdiff(out[2], g["gate"]) < TOL
    for i in range(cfg["n_flows"]):
        assert _maxdiff(out[1][i], g["log_s"][i]) < TOL
        assert _maxdiff(out[3][i], g["attn"][i]) < TOL
        assert _maxdiff(out[4][i], g["logprob"][i]) < 2e-4      # log of small probabilities
    nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    assert abs(nll.item() - g["nll"].item()) < 1e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 1e-5
    assert abs(ctc.item() - g["ctc"].item()) < 1e-4
    (nll + gl + 0.01 * ctc).sum().backward()
    for k, ref in g["grads"].items():
        mine = sd[k].grad
        assert mine is not None, k
        # conv biases ahead of an instance norm have an analytically ZERO gradient (the norm
        # removes the mean): both sides hold fp32 round-off there, so floor the denominator
        denom = max(ref.norm().item(), 1e-5 * ref.numel() ** 0.5)
        assert (mine - ref).norm().item() / denom < 2e-4, (k, (mine - ref).norm().item() / denom)


@pytest.mark.parametrize("name", ["small_f2.pt", "small_f3.pt", "small_cumm.pt", "small_dummy_spk.pt"])
def test_infer_small(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = g["cfg"]
    sd = synth.make_state_dict(cfg, seed=g["seed"])
    b = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=g["with_prior"])
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, cfg["n_mel_channels"], n)).astype(np.float32)) * 0.5
    txt, spk = b["text"][:1, : g["in_lens"][0]], b["speaker_ids"][:1]
    mel, attns = O.infer(sd, cfg, residual, spk, txt, gate_threshold=1.0)
    assert _maxdiff(mel, g["infer_mel"]) < TOL
    for a, ra in zip(attns, g["infer_attn"]):
        assert _maxdiff(a, ra) < TOL
    mel_g, _ = O.infer(sd, cfg, residual, spk, txt, gate_threshold=0.5)
    assert mel_g.shape[2] == g["infer_gated_frames"]
    if "infer_prior_mel" in g:
        pr = O.beta_binomial_prior(g["in_lens"][0], n).float()[None]
        mel_p, attns_p = O.infer(sd, cfg, residual, spk, txt, gate_threshold=1.0, attn_prior=pr)
        assert _maxdiff(mel_p, g["infer_prior_mel"]) < TOL
        for a, ra in zip(attns_p, g["infer_prior_attn"]):
            assert _maxdiff(a, ra) < TOL
        # forced alignment: feeding a free run's own attention back reproduces it (flows order = reversed output order)
        mel_f, _ = O.infer(sd, cfg, residual, spk, txt, gate_threshold=1.0, attns=attns[::-1])
        assert _maxdiff(mel_f, mel) < 1e-6


def test_cfg1_full_size(golden_dir):
    """BASELINE config 1: 1-flow, n_text=148, B=2, T=800/650, fp32 (forward + losses + infer)."""
    g = _load(golden_dir, "cfg1_full.pt")
    cfg = g["cfg"]
    sd = synth.make_state_dict(cfg, seed=g["seed"])
    b = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=True)
    O.LSTM_IMPL["fn"] = O.lstm_seq_fast
    try:
        with torch.no_grad():
            out = O.forward(sd, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    finally:
        O.LSTM_IMPL["fn"] = O.lstm_cell_seq
    st = g["stride"]
    assert _maxdiff(out[0][::st], g["z"]) < 1e-4
    assert _maxdiff(out[1][0][::st], g["log_s"][0]) < 1e-4
    assert _maxdiff(out[3][0][:, ::st], g["attn"][0]) < 1e-5
    assert abs(nll.item() - g["nll"].item()) < 1e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 1e-5
    assert abs(ctc.item() - g["ctc"].item()) < 1e-4 * max(1.0, abs(g["ctc"].item()))
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, 80, n)).astype(np.float32)) * 0.5
    with torch.no_grad():
        mel, _ = O.infer(sd, cfg, residual, b["speaker_ids"][:1], b["text"][:1, : g["in_lens"][0]], gate_threshold=1.0)
    assert _maxdiff(mel, g["infer_mel"]) < 1e-4


def test_stft_mel(golden_dir):
    g = _load(golden_dir, "stft_mel.pt")
    y = torch.stack([synth.make_audio(g["n_samples"], seed=s) for s in g["seeds"]])
    mel = O.stft_mel(y)
    assert mel.shape == g["mel"].shape
    assert _maxdiff(mel, g["mel"]) < 2e-4


def test_beta_binomial_prior(golden_dir):
    g = _load(golden_dir, "prior.pt")
    assert _maxdiff(O.beta_binomial_prior(13, 40), g["p13_m40"]) < 1e-11
    assert _maxdiff(O.beta_binomial_prior(148, 800)[::50], g["p148_m800_s50"]) < 1e-11


def test_reverse_by_length_is_flip_roll():
    torch.manual_seed(0)
    x = torch.randn(9, 3, 4)
    lens = torch.tensor([9, 4, 1])
    y = O.reverse_by_length(x, lens, 0, 1)
    ref = torch.flip(x, (0,)).clone()
    for k in range(3):
        ref[:, k] = ref[:, k].roll(int(lens[k]), dims=0)          # flowtron.py:606-613
    assert torch.equal(y, ref)
    assert torch.equal(O.reverse_by_length(y, lens, 0, 1), x)     # involution


def test_libritts_full_width_oracle_vs_real_reference(golden_dir):
    """cfg_libritts.pt: the REAL reference on the LibriTTS-shaped full-width case (tests/libri_case.py: 123 speakers, L = 237,
    ragged B = 4, H = 1024, prior + CTC): the oracle's losses within 1e-5 rel and every parameter's (sampled) gradient within
    2e-4 rel-L2 -- this is the pin of the oracle at BASELINE configs[2] / configs[4], the shape the GPU test then checks the
    HIP path against."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import libri_case
    g = _load(golden_dir, "cfg_libritts.pt")
    cfg, sd, b = libri_case.make()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    O.LSTM_IMPL["fn"] = O.lstm_seq_fast
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(sdg, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    (nll + gl + 0.01 * ctc).sum().backward()
    rn, rg, rc = (x.item() for x in g["losses_fp32"])
    assert abs(nll.item() - rn) < 1e-5 * abs(rn) and abs(gl.item() - rg) < 1e-5 and abs(ctc.item() - rc) < 1e-4 * abs(rc)
    worst = ("", 0.0)
    for k, e in g["grad"].items():
        mine = sdg[k].grad.reshape(-1)
        mine = mine if e["idx"] is None else mine[e["idx"]]
        if k.startswith("encoder.convolutions") and k.endswith("conv.bias"):      # exact value 0 (instance norm removes the constant)
            assert mine.abs().max().item() < 1e-6 and e["sample"].abs().max().item() < 1e-6
            continue
        r = (mine - e["sample"]).norm().item() / max(e["sample"].norm().item(), 1e-30)
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] < 2e-4, worst


def test_chunked_oracle_equals_one_shot_oracle():
    """tests/oracle_chunked.py (the oracle over a big batch, a few utterances at a time -- what the full-size GPU parity test
    of BASELINE configs[1] uses) returns the one-shot oracle's losses and gradients: utterances are independent and every loss
    term is a weighted sum over them."""
    import oracle_chunked as OC
    cfg = dict(synth.SMALL_MODEL_CONFIG)
    out_lens, in_lens = [23, 19, 17, 12, 9, 5], [9, 8, 8, 6, 4, 3]
    sd = synth.make_state_dict(cfg, seed=5)
    b = synth.make_batch(cfg, out_lens, in_lens, seed=5, with_prior=True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(sdg, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    (nll + gl + 0.01 * ctc).sum().backward()
    for chunk in (2, 4):
        (n2, g2, c2), grads = OC.forward_backward(cfg, sd, b, b["attn_prior"], chunk=chunk)
        assert abs(n2 - nll.item()) < 1e-5 * abs(nll.item()) and abs(g2 - gl.item()) < 1e-5 and abs(c2 - ctc.item()) < 1e-4
        for k, v in sdg.items():
            denom = max(v.grad.norm().item(), 1e-5 * v.numel() ** 0.5)
            assert (grads[k] - v.grad).norm().item() / denom < 2e-4, (k, chunk)


@pytest.mark.parametrize("T,B,I,H,reverse", [(13, 5, 24, 16, False), (13, 5, 24, 16, True), (40, 3, 8, 32, False), (7, 1, 8, 8, True)])
def test_lstm_variants_equal_the_explicit_recurrence(T, B, I, H, reverse):
    """The oracle's three statements of the length-masked LSTM (flowtron.py:689-694) agree on ragged batches, outputs and every
    gradient: the explicit recurrence (the definition, pinned by the goldens), torch's packed CPU LSTM (what the reference
    executes) and torch's CPU LSTM over the padded batch with pad frames zeroed (what the T = 862 GPU parity test and bench.py's
    cpu_baseline use: the packed path's autograd is O(T^2))."""
    torch.manual_seed(T + H)
    lens = torch.tensor(sorted([T] + [int(v) for v in torch.randint(1, T + 1, (B - 1,))], reverse=True))
    x = torch.randn(T, B, I)
    ws = [torch.randn(4 * H, I) * 0.3, torch.randn(4 * H, H) * 0.3, torch.randn(4 * H) * 0.1, torch.randn(4 * H) * 0.1]
    g = torch.randn(T, B, H)
    res = []
    for fn in (O.lstm_cell_seq, O.lstm_seq_fast, O.lstm_seq_padded):
        xx = x.clone().requires_grad_(True)
        w = [v.clone().requires_grad_(True) for v in ws]
        y = fn(xx, lens, *w, reverse=reverse)
        (y * g).sum().backward()
        res.append([y.detach(), xx.grad] + [v.grad for v in w])
    for other in res[1:]:
        for a, b_ in zip(res[0], other):
            assert _maxdiff(a, b_) < 1e-5
    pad = ~(torch.arange(T)[:, None] < lens[None, :])
    assert float(res[2][0][pad].abs().max() if pad.any() else 0.0) == 0.0      # pad frames: exactly zero, like pad_packed_sequence


def test_chunked_oracle_on_the_padded_lstm_equals_one_shot_oracle():
    """the combination the full-size GPU parity test uses: oracle_chunked + O.lstm_seq_padded == the one-shot oracle on the
    explicit recurrence"""
    import oracle_chunked as OC
    cfg = dict(synth.SMALL_MODEL_CONFIG)
    out_lens, in_lens = [23, 19, 17, 12, 9, 5], [9, 8, 8, 6, 4, 3]
    sd = synth.make_state_dict(cfg, seed=6)
    b = synth.make_batch(cfg, out_lens, in_lens, seed=6, with_prior=True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(sdg, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    (nll + gl + 0.01 * ctc).sum().backward()
    O.LSTM_IMPL["fn"] = O.lstm_seq_padded
    try:
        (n2, g2, c2), grads = OC.forward_backward(cfg, sd, b, b["attn_prior"], chunk=4)
    finally:
        O.LSTM_IMPL["fn"] = O.lstm_cell_seq
    assert abs(n2 - nll.item()) < 1e-5 * abs(nll.item()) and abs(g2 - gl.item()) < 1e-5 and abs(c2 - ctc.item()) < 1e-4
    for k, v in sdg.items():
        denom = max(v.grad.norm().item(), 1e-5 * v.numel() ** 0.5)
        assert (grads[k] - v.grad).norm().item() / denom < 2e-4, k


def test_cumulative_attention_full_width_oracle_vs_real_reference(golden_dir):
    """the oracle's location-sensitive attention branch (attn_cond / cumm_attention_sequence / the infer loop) at FULL width
    (H 1024, A 640, T 400) against the real reference's golden (cumm_full.pt): forward outputs, losses, 48-frame inference."""
    g = _load(golden_dir, "cumm_full.pt")
    cfg = g["cfg"]
    sd = synth.make_state_dict(cfg, seed=g["seed"])
    b = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=True)
    O.LSTM_IMPL["fn"] = O.lstm_seq_fast
    try:
        with torch.no_grad():
            out = O.forward(sd, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
    finally:
        O.LSTM_IMPL["fn"] = O.lstm_cell_seq
    st = g["stride"]
    assert _maxdiff(out[0][::st], g["z"]) < 2e-4
    for i in range(2):
        assert _maxdiff(out[1][i][::st], g["log_s"][i]) < 2e-4
        assert _maxdiff(out[3][i][:, ::st], g["attn"][i]) < 2e-5
    assert abs(nll.item() - g["nll"].item()) < 1e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 1e-5
    assert abs(ctc.item() - g["ctc"].item()) < 1e-4 * abs(g["ctc"].item())
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, 80, n)).astype(np.float32)) * 0.5
    with torch.no_grad():
        mel, attn = O.infer(sd, cfg, residual, b["speaker_ids"][:1], b["text"][:1, : g["in_lens"][0]], gate_threshold=1.0)
    assert _maxdiff(mel, g["infer_mel"]) < 1e-4


def test_training_trajectory_oracle_vs_real_reference(golden_dir):
    """five iterations of train.py:282-331 (zero_grad, forward, loss, backward, clip_grad_norm_, RAdam step) by the
    REAL reference modules on CPU (tests/golden/make_golden_r4.py -> train_traj.pt) against the oracle's forward / loss with
    autograd, O.clip_grad_norm and O.radam_step: losses and pre-clip gradient norms of every iteration, every parameter at the end."""
    g = _load(golden_dir, "train_traj.pt")
    spec = g["spec"]
    cfg = spec["cfg"]
    params = {k: v.clone() for k, v in synth.make_state_dict(cfg, seed=spec["seed"]).items()}
    batches = [synth.make_batch(cfg, b["out_lens"], b["in_lens"], seed=b["seed"], with_prior=True) for b in spec["batches"]]
    state = {}
    for it in range(spec["iters"]):
        b = batches[it % len(batches)]
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        out = O.forward(sd, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
        nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], spec["sigma"], True, True, spec["blank_logprob"])
        loss = nll + gl + spec["ctc_loss_weight"] * ctc
        ref = g["losses"][it]
        for mine, r in zip((loss, gl, nll, ctc), ref.tolist()):
            assert abs(float(mine) - r) < 2e-5 * max(1.0, abs(r)), (it, float(mine), r)
        loss.sum().backward()
        total, grads = O.clip_grad_norm({k: v.grad for k, v in sd.items()}, spec["grad_clip_val"])
        assert abs(float(total) - float(g["grad_norms"][it])) < 1e-4 * float(g["grad_norms"][it]), (it, float(total))
        with torch.no_grad():
            O.radam_step(params, grads, state, lr=spec["lr"], weight_decay=spec["weight_decay"])
    assert g["optimizer_steps"] == [spec["iters"]]
    init = synth.make_state_dict(cfg, seed=spec["seed"])
    for k, ref in g["params"].items():
        moved = (ref - init[k]).abs().max().item()
        assert _maxdiff(params[k], ref) < 2e-6 + 2e-3 * moved, (k, _maxdiff(params[k], ref), moved)


def test_decoder_lstm_depths_one_and_three_vs_real_reference(golden_dir):
    """n_lstm_layers is a config.json key that flowtron.py:655 hands to nn.LSTM: the oracle at depth 1 and 3 against the real reference
    (tests/golden/make_golden_r4.py -> lstm_depth.pt): z, the three losses, every gradient."""
    g = _load(golden_dir, "lstm_depth.pt")
    for c in g["cases"]:
        case, cfg = c["case"], c["cfg"]
        sd = {k: v.clone().requires_grad_(True) for k, v in synth.make_state_dict(cfg, seed=case["seed"]).items()}
        assert ("flows.0.lstm.weight_ih_l%d" % (case["n_lstm_layers"] - 1)) in sd and ("flows.0.lstm.weight_ih_l%d" % case["n_lstm_layers"]) not in sd
        b = synth.make_batch(cfg, case["out_lens"], case["in_lens"], seed=case["seed"], with_prior=True)
        out = O.forward(sd, cfg, b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
        assert _maxdiff(out[0], c["z"]) < TOL
        nll, gl, ctc = O.loss(out, b["gate_target"], b["in_lens"], b["out_lens"], 1.0, True, True, -8)
        assert abs(nll.item() - c["nll"].item()) < 1e-5 * abs(c["nll"].item()) and abs(gl.item() - c["gate_loss"].item()) < 1e-5
        assert abs(ctc.item() - c["ctc"].item()) < 1e-4
        (nll + gl + 0.01 * ctc).sum().backward()
        assert set(c["grads"]) == set(sd)
        for k, ref in c["grads"].items():
            denom = max(ref.norm().item(), 1e-5 * ref.numel() ** 0.5)
            assert (sd[k].grad - ref).norm().item() / denom < 2e-4, (case["n_lstm_layers"], k)


def test_infer_depth_and_batch_vs_reference_golden(golden_dir):
    """Flowtron.infer of the REAL reference (tests/golden/infer_depth.pt, make_golden_r5.py) at decoder depths 1 and 3
    (flowtron.py:654-655) and for a batch of two utterances (:775-828 takes any batch when there is no gate layer): the oracle's
    infer loop follows it -- mel, the attention rows of both flows, the gated frame count."""
    sys.path.insert(0, golden_dir)
    import make_golden_r5 as G5
    g = _load(golden_dir, "infer_depth.pt")
    for ref in g["cases"]:
        case = ref["case"]
        cfg, sd, residual, spk, text = G5.case_inputs(case)
        mel, attns = O.infer(sd, cfg, residual, spk, text, gate_threshold=1.0)
        assert mel.shape == ref["mel"].shape and _maxdiff(mel, ref["mel"]) < TOL, case["name"]
        for a, ra in zip(attns, ref["attn"]):                      # decode order; reference rows [B, N, L]
            a = a[None] if a.dim() == 2 else a.permute(1, 0, 2)
            assert _maxdiff(a, ra) < TOL, case["name"]
        if "gated_frames" in ref:
            mel_g, _ = O.infer(sd, cfg, residual, spk, text, gate_threshold=0.5)
            assert mel_g.shape[2] == ref["gated_frames"], case["name"]
