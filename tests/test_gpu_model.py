"""Model-level parity (-m gpu): flowtron.Flowtron / FlowtronLoss (HIP path through the C ABI) against
 (a) the committed golden vectors that tests/golden/make_golden.py produced by running the REAL reference, and
 (b) the CPU oracle on fresh seeded inputs, plus size-independent properties (invertibility, padding
     invariance) at the full 2-flow config.  fp32 tolerances follow SURVEY 8c."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def mad(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def build(cfg, seed, mode="f32"):
    import flowtron
    from oracle import synth
    os.environ["FLOWTRON_MFMA"] = mode
    m = flowtron.Flowtron(**cfg)
    sd = synth.make_state_dict(cfg, seed=seed)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


def cuda_batch(b):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}


@pytest.mark.parametrize("name", ["small_f2.pt", "small_f3.pt", "small_cumm.pt", "small_dummy_spk.pt"])
def test_forward_loss_grads_vs_reference_golden(name):
    import flowtron
    from oracle import synth
    g = _load(name)
    cfg = g["cfg"]
    m, _ = build(cfg, g["seed"])
    b = cuda_batch(synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=g["with_prior"]))
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    assert mad(out[0], g["z"]) < 1e-4
    assert mad(out[2], g["gate"]) < 1e-4
    for i in range(cfg["n_flows"]):
        assert mad(out[1][i], g["log_s"][i]) < 1e-4, i
        assert mad(out[3][i], g["attn"][i]) < 1e-5, i
        assert mad(out[4][i], g["logprob"][i]) < 5e-4, i
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    assert abs(nll.item() - g["nll"].item()) < 1e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 1e-5
    assert abs(ctc.item() - g["ctc"].item()) < 1e-4 * max(1.0, abs(g["ctc"].item()))
    (nll + gl + 0.01 * ctc).sum().backward()
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        assert p.grad is not None, k
        denom = max(ref.norm().item(), 1e-5 * ref.numel() ** 0.5)
        r = (p.grad.cpu() - ref).norm().item() / denom
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] < 1e-4, worst          # SURVEY 8c's fp32 gradient tolerance (observed ~2e-5)


@pytest.mark.parametrize("name", ["small_f2.pt", "small_f3.pt", "small_cumm.pt", "small_dummy_spk.pt"])
@pytest.mark.parametrize("use_graph", ["0", "1"])
def test_infer_vs_reference_golden(name, use_graph):
    from oracle import synth
    os.environ["FLOWTRON_DECODE_GRAPH"] = use_graph
    g = _load(name)
    cfg = g["cfg"]
    m, _ = build(cfg, g["seed"])
    b = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=g["with_prior"])
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, cfg["n_mel_channels"], n)).astype(np.float32)) * 0.5
    txt, spk = b["text"][:1, : g["in_lens"][0]].cuda(), b["speaker_ids"][:1].cuda()
    mel, attns = m.infer(residual.cuda(), spk, txt, gate_threshold=1.0)
    assert mel.shape == g["infer_mel"].shape
    assert mad(mel, g["infer_mel"]) < 1e-4
    for a, ra in zip(attns, g["infer_attn"]):
        assert mad(torch.cat(a)[:, 0], ra) < 1e-5
    mel_g, _ = m.infer(residual.cuda(), spk, txt, gate_threshold=0.5)
    assert mel_g.shape[2] == g["infer_gated_frames"]
    if "infer_prior_mel" in g:                          # attention prior at inference (flowtron.py:799), real-reference golden
        from oracle import flowtron_oracle as O
        pr = O.beta_binomial_prior(g["in_lens"][0], n).float()[None].cuda()
        mel_p, attns_p = m.infer(residual.cuda(), spk, txt, gate_threshold=1.0, attn_prior=pr)
        assert mad(mel_p, g["infer_prior_mel"]) < 1e-4
        for a, ra in zip(attns_p, g["infer_prior_attn"]):
            assert mad(torch.cat(a)[:, 0], ra) < 1e-5
        # forced alignment (flowtron.py:798): a free run's own attention rows, given back per flow, reproduce its mel
        forced = [torch.cat(a)[:, 0] for a in attns][::-1]
        mel_f, _ = m.infer(residual.cuda(), spk, txt, gate_threshold=1.0, attns=forced)
        assert mad(mel_f, mel) < 1e-5


@pytest.mark.parametrize("use_graph", ["0", "1"])
def test_infer_depth_and_batch_vs_reference_golden(use_graph):
    """Decode breadth: Flowtron.infer at decoder depths 1 and 3 (flowtron.py:654-655; staged chain: a layer beyond
    the second is stage 5 again with its own weights and state) and for a batch of two utterances (flowtron.py:775-828; decoded one
    after the other through the batch-1 kernels) against golden vectors of the REAL reference (tests/golden/infer_depth.pt,
    make_golden_r5.py): mel 1e-4, attention rows 1e-5, the gated frame count.  fp32 operand mode, hipGraph on and off."""
    import sys
    import flowtron
    sys.path.insert(0, GOLDEN)
    import make_golden_r5 as G5
    os.environ["FLOWTRON_DECODE_GRAPH"] = use_graph
    os.environ["FLOWTRON_MFMA"] = "f32"
    g = _load("infer_depth.pt")
    for ref in g["cases"]:
        case = ref["case"]
        cfg, sd, residual, spk, text = G5.case_inputs(case)
        m = flowtron.Flowtron(**cfg)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        mel, attns = m.infer(residual.cuda(), spk.cuda(), text.cuda(), gate_threshold=1.0)
        assert mel.shape == ref["mel"].shape, case["name"]
        assert mad(mel, ref["mel"]) < 1e-4, (case["name"], mad(mel, ref["mel"]))
        for a, ra in zip(attns, ref["attn"]):                  # decode order; ours: N rows of [B,1,L], reference [B,N,L]
            assert mad(torch.cat(a, 1), ra) < 1e-5, case["name"]
        if "gated_frames" in ref:
            mel_g, _ = m.infer(residual.cuda(), spk.cuda(), text.cuda(), gate_threshold=0.5)
            assert mel_g.shape[2] == ref["gated_frames"], case["name"]
    # 16-bit operand mode at depth 3: the layers beyond the second stream fp32 weights beside the images of the rest
    os.environ["FLOWTRON_MFMA"] = "bf16"
    ref = g["cases"][1]
    cfg, sd, residual, spk, text = G5.case_inputs(ref["case"])
    m = flowtron.Flowtron(**cfg)
    m.load_state_dict(sd)
    mel, _ = m.cuda().eval().infer(residual.cuda(), spk.cuda(), text.cuda(), gate_threshold=1.0)
    os.environ["FLOWTRON_MFMA"] = "f32"
    assert mad(mel, ref["mel"]) < 5e-2, mad(mel, ref["mel"])


def test_gated_batch_decode_equals_the_utterances_decoded_alone():
    """with a gate layer every utterance of a batch stops at its OWN frame of the last flow, and the flows decoded after it
    must see exactly that utterance's frames (flowtron.py:775-828 per utterance; 901-930).  A batch of three against the same three
    utterances decoded alone, at a gate threshold where the stops differ: mel and attention rows identical on each utterance's frames,
    zero behind its stop."""
    import flowtron
    from oracle import synth
    os.environ["FLOWTRON_MFMA"] = "f32"
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60, n_flows=2)
    sd = synth.make_state_dict(cfg, seed=17)
    gk = [k for k in sd if "gate_layer" in k and k.endswith("weight")][0]
    torch.manual_seed(4)
    sd[gk] = torch.randn_like(sd[gk]) * 0.05               # gate logits of about one unit spread around the bias: stops at different frames
    sd[gk.replace("weight", "bias")] = torch.full_like(sd[gk.replace("weight", "bias")], -2.0)
    m = flowtron.Flowtron(**cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    B, N, Lk = 3, 90, 21
    residual = (torch.randn(B, cfg["n_mel_channels"], N) * 0.5).cuda()
    text = torch.randint(1, 60, (B, Lk)).cuda()
    spk = torch.zeros(B, dtype=torch.long).cuda()
    found = False
    for thr in (0.03, 0.05, 0.08, 0.12, 0.16, 0.2, 0.25, 0.3, 0.4, 0.5):
        alone = [m.infer(residual[b:b + 1], spk[b:b + 1], text[b:b + 1], gate_threshold=thr) for b in range(B)]
        lens = [int(a[0].shape[2]) for a in alone]
        if len(set(lens)) > 1 and min(lens) < N:
            found = True
            break
    assert found, "no threshold separated the stops: %r" % (lens,)
    mel, attns = m.infer(residual, spk, text, gate_threshold=thr)
    assert mel.shape[2] == max(lens)
    for b in range(B):
        assert torch.equal(mel[b, :, :lens[b]], alone[b][0][0]), (b, lens)
        assert float(mel[b, :, lens[b]:].abs().max()) == 0.0 if lens[b] < max(lens) else True
        for f in range(cfg["n_flows"]):
            rows_b = torch.cat([r[b] for r in attns[f][:lens[b]]], 0)
            rows_a = torch.cat([r[0] for r in alone[b][1][f]], 0)
            assert torch.equal(rows_b, rows_a), (b, f)


def test_infer_bf16_weight_images_and_persistent_decode_track_fp32_full_width():
    """bf16 operand mode decodes from bf16 IMAGES of the weights, by default as ONE persistent launch per flow (csrc/decode.hip
    dec_persist_k: 256 workgroups hand every stage vector to one another through tag-checked granules), else as the staged
    launch chain (one workgroup per hidden unit, wave per gate row).  Full-width 2-flow model, 48 frames: both against the
    fp32-weight decode of the parity mode (which the real-reference goldens pin) -- weight rounding 2^-9 relative through
    ~100 recurrent steps: mel within 5e-2 (values span ~[-12, 2]), attention rows within 2e-2 -- and against each other
    (same bf16 weights, different summation order: 2e-3).  The gate stop count must agree between all three."""
    from flowtron_amd import ops
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60)
    rs = np.random.RandomState(5)
    residual = torch.from_numpy(rs.standard_normal((1, 80, 48)).astype(np.float32)).cuda() * 0.5
    txt = torch.from_numpy(rs.randint(0, 60, (1, 23))).cuda()
    spk = torch.zeros(1, dtype=torch.long).cuda()
    res = {}
    try:
        for name, mode, persist in (("f32", "f32", "0"), ("staged", "bf16", "0"), ("persist", "bf16", "1")):
            os.environ["FLOWTRON_DECODE_PERSIST"] = persist
            m, _ = build(cfg, 17, mode)
            res[name] = m.infer(residual, spk, txt, gate_threshold=1.0)
            res[name + "_gated"] = m.infer(residual, spk, txt, gate_threshold=0.45)[0].shape[2]
            ops.check_persist_status()
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
        os.environ.pop("FLOWTRON_DECODE_PERSIST", None)
    for name in ("staged", "persist"):
        assert res[name][0].shape == res["f32"][0].shape == (1, 80, 48)
        assert torch.isfinite(res[name][0]).all()
        assert mad(res[name][0], res["f32"][0]) < 5e-2, (name, mad(res[name][0], res["f32"][0]))
        for a16, a32 in zip(res[name][1], res["f32"][1]):
            assert mad(torch.cat(a16), torch.cat(a32)) < 2e-2
    assert mad(res["persist"][0], res["staged"][0]) < 2e-3, mad(res["persist"][0], res["staged"][0])
    assert res["persist_gated"] == res["staged_gated"] == res["f32_gated"]


def test_decode_400_frames_vs_oracle_all_three_decoders(capsys):   # (four since round 4: + the fp32 persistent decode)
    """BASELINE configs[3] at the shape bench.py times (2-flow LJS model, B = 1, L = 69 text symbols, 400 residual frames,
    sigma = 0.5) against the fp32 CPU ORACLE (O.infer = restatement of flowtron.py:775-828, 901-930), for every decoder the
    library has: the fp32-weight staged hipGraph chain (the reference's own arithmetic, inference.py:68-71), the fp32 one-launch
    persistent decode (round 4: dec_persist_k<true>, recurrent matrices register-resident, the rest streamed from the L2), the
    bf16-image staged chain and the bf16 one-launch persistent decode (dec_persist_k<false>, the one behind the headline RTF).
    Tolerances on mel (values span ~[-2, 2] here) and attention rows (probabilities) over all 400 frames x 2 flows, ~10x what the
    MI355X measures (run A of round 3: fp32 3.6e-7 / 1.9e-8; bf16 staged and persistent 8.6e-5 max, 1.4e-5 mean / 3.2e-5):
      fp32 weights : mel 1e-5, attention 1e-6 (400 sequentially dependent frames of fp32 re-association);
      bf16 weights : mel 1e-3 max / 2e-4 mean, attention 5e-4 (weights rounded to 8 significand bits, 800 recurrent steps).
    Gate: random weights give a gate that hovers at 0.50-0.53 with no usable margin, so the test DESIGNS the gate layer from the
    oracle's own trajectory -- the minimum-norm weight with logit -1 on frames 0 .. 249 and +1 on frame 250 of the gated flow
    (sigmoid 0.27 / 0.73 around the threshold 0.5; a 3 % perturbation of the gate input moves the logit by 0.03) -- and every
    decoder must stop at exactly the oracle's frame."""
    import flowtron
    from flowtron_amd import ops
    from oracle import flowtron_oracle as O
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    N, Lk, f_stop = 400, 69, 250
    rs = np.random.RandomState(404)
    residual = torch.from_numpy(rs.standard_normal((1, 80, N)).astype(np.float32)) * 0.5
    txt = torch.from_numpy(rs.randint(0, cfg["n_text"], (1, Lk)))
    spk = torch.zeros(1, dtype=torch.long)
    sd = synth.make_state_dict(cfg, seed=23)
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    gates = []
    with torch.no_grad():
        ref_mel, ref_attn = O.infer(sd, cfg, residual, spk, txt, gate_threshold=2.0, gates_out=gates)
    assert ref_mel.shape == (1, 80, N) and len(gates) == N
    D = torch.stack([g[1] for g in gates]).double()                       # [N, H + A] gate inputs of the gated (last) flow
    y = -torch.ones(f_stop + 1, dtype=torch.float64)
    y[f_stop] = 1.0
    w = torch.linalg.pinv(D[: f_stop + 1]) @ y
    gk = [k for k in sd if k.endswith("gate_layer.linear_layer.weight")][0]
    sd[gk] = w.float().reshape(1, -1)
    sd[gk.replace("weight", "bias")] = torch.zeros(1)
    with torch.no_grad():
        assert O.infer(sd, cfg, residual, spk, txt, gate_threshold=0.5)[0].shape[2] == f_stop + 1
    res = {}
    try:
        for name, mode, persist in (("f32", "f32", "0"), ("f32_persist", "f32", "1"), ("bf16_staged", "bf16", "0"), ("bf16_persist", "bf16", "1")):
            os.environ["FLOWTRON_DECODE_PERSIST"] = persist
            os.environ["FLOWTRON_MFMA"] = mode
            m = flowtron.Flowtron(**cfg)
            m.load_state_dict(sd)
            m = m.cuda().eval()
            mel, attns = m.infer(residual.cuda(), spk.cuda(), txt.cuda(), gate_threshold=2.0)
            n_gated = m.infer(residual.cuda(), spk.cuda(), txt.cuda(), gate_threshold=0.5)[0].shape[2]
            ops.check_persist_status()
            res[name] = (mel.cpu(), [torch.cat(a)[:, 0].cpu() for a in attns], n_gated)
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
        os.environ.pop("FLOWTRON_DECODE_PERSIST", None)
    rows = []
    for name, (mel, attns, n_gated) in res.items():
        assert mel.shape == ref_mel.shape and torch.isfinite(mel).all()
        d = (mel - ref_mel).abs()
        da = max(mad(a, r) for a, r in zip(attns, ref_attn))
        rows.append((name, d.max().item(), d.mean().item(), da, n_gated))
    with capsys.disabled():
        print("\n[decode 400 frames x 2 flows vs fp32 oracle] oracle stops at frame %d; |mel| <= %.2f" % (f_stop + 1, ref_mel.abs().max().item()))
        for r in rows:
            print("   %-13s mel max %.2e mean %.2e | attention max %.2e | gated frames %d" % r)
    for name, dmax, dmean, da, n_gated in rows:
        if name.startswith("f32"):
            assert dmax < 1e-5 and da < 1e-6, (name, dmax, da)
        else:
            assert dmax < 1e-3 and dmean < 2e-4 and da < 5e-4, (name, dmax, dmean, da)
        assert n_gated == f_stop + 1, (name, n_gated, f_stop + 1)


def test_cfg1_full_size_vs_reference_golden():
    """BASELINE config 1: 1-flow, n_text=148, B=2, T=800/650, L=148/120, fp32, prior + CTC on."""
    import flowtron
    from oracle import synth
    g = _load("cfg1_full.pt")
    cfg = g["cfg"]
    m, _ = build(cfg, g["seed"])
    b = cuda_batch(synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=True))
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    st = g["stride"]
    assert mad(out[0][::st], g["z"]) < 2e-4
    assert mad(out[1][0][::st], g["log_s"][0]) < 2e-4
    assert mad(out[3][0][:, ::st], g["attn"][0]) < 2e-5
    assert abs(nll.item() - g["nll"].item()) < 2e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 2e-5
    assert abs(ctc.item() - g["ctc"].item()) < 2e-4 * max(1.0, abs(g["ctc"].item()))
    (nll + gl + 0.01 * ctc).sum().backward()
    bad = []
    for k, p in m.named_parameters():
        ref_n = g["grad_norm"][k]
        samp = g["grad_sample"][k]
        mine = p.grad.cpu().flatten()[:: max(1, p.numel() // 64)][:64]
        scale = max(ref_n / p.numel() ** 0.5, 1e-7)
        if abs(p.grad.norm().item() - ref_n) > 2e-3 * max(ref_n, 1e-5 * p.numel() ** 0.5) or (mine - samp).abs().max().item() > 2e-2 * scale + 1e-7:
            bad.append((k, p.grad.norm().item(), ref_n, (mine - samp).abs().max().item(), scale))
    assert not bad, bad[:5]
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, 80, n)).astype(np.float32)) * 0.5
    mel, _ = m.infer(residual.cuda(), b["speaker_ids"][:1], b["text"][:1, : g["in_lens"][0]], gate_threshold=1.0)
    assert mad(mel, g["infer_mel"]) < 2e-4


def test_cumulative_attention_full_width_vs_reference_golden(capsys):
    """SURVEY 8a row a17 at FULL width: use_cumm_attention = True (location-sensitive attention: Conv1d(2->32,k5) + ReLU +
    Conv1d(32->640,k3) + sigmoid over [cumulative ; previous] attention modulating the keys, flowtron.py:129-152, 697-723, 793-806),
    H 1024 / A 640 / E 640, 2 flows, B = 2, T = 400 / 333 -- against golden vectors of the REAL reference
    (tests/golden/cumm_full.pt, make_golden_r3.py --cumm): strided forward outputs, the three losses, gradient norms + samples of
    all 76 parameter tensors, and a 48-frame inference.  fp32 MFMA mode; tolerances as test_cfg1_full_size_vs_reference_golden
    (400 frames of a feedback loop through the attention: 5e-4 on z / log_s)."""
    import time
    import flowtron
    from oracle import synth
    g = _load("cumm_full.pt")
    cfg = g["cfg"]
    m, _ = build(cfg, g["seed"])
    b = cuda_batch(synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).sum().backward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = g["stride"]
    dz = mad(out[0][::st], g["z"])
    dls = max(mad(out[1][i][::st], g["log_s"][i]) for i in range(2))
    dat = max(mad(out[3][i][:, ::st], g["attn"][i]) for i in range(2))
    with capsys.disabled():
        print("\n[cumulative attention, full width, T 400, B 2] fwd+loss+bwd %.2f s (first call) | z %.2e log_s %.2e attn %.2e | nll %.6f / %.6f ctc %.5f / %.5f"
              % (dt, dz, dls, dat, nll.item(), g["nll"].item(), ctc.item(), g["ctc"].item()))
    assert dz < 5e-4 and dls < 5e-4 and dat < 5e-5, (dz, dls, dat)
    assert mad(out[2][::st], g["gate"]) < 5e-4
    assert abs(nll.item() - g["nll"].item()) < 2e-5 * abs(g["nll"].item())
    assert abs(gl.item() - g["gate_loss"].item()) < 2e-5
    assert abs(ctc.item() - g["ctc"].item()) < 2e-4 * max(1.0, abs(g["ctc"].item()))
    bad = []
    for k, p in m.named_parameters():
        ref_n = g["grad_norm"][k]
        samp = g["grad_sample"][k]
        mine = p.grad.cpu().flatten()[:: max(1, p.numel() // 64)][:64]
        scale = max(ref_n / p.numel() ** 0.5, 1e-7)
        if abs(p.grad.norm().item() - ref_n) > 5e-3 * max(ref_n, 1e-5 * p.numel() ** 0.5) or (mine - samp).abs().max().item() > 5e-2 * scale + 1e-7:
            bad.append((k, p.grad.norm().item(), ref_n, (mine - samp).abs().max().item(), scale))
    assert not bad, bad[:5]
    n = g["infer_mel"].shape[2]
    rs = np.random.RandomState(g["seed"] + 11)
    residual = torch.from_numpy(rs.standard_normal((1, 80, n)).astype(np.float32)) * 0.5
    mel, attns = m.infer(residual.cuda(), b["speaker_ids"][:1], b["text"][:1, : g["in_lens"][0]], gate_threshold=1.0)
    assert mad(mel, g["infer_mel"]) < 2e-4
    for a, ra in zip(attns, g["infer_attn"]):
        assert mad(torch.cat(a)[:, 0], ra) < 2e-5


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_cumulative_attention_library_walk_equals_python_walk(mode, tol):
    """csrc/cumm_attn.hip (the library walks the frames: ops.CummAttnSeqFn) against the per-frame autograd walk it replaces
    (FLOWTRON_CUMM_LOOP=python: the same im2col / GEMM / activation kernels, the unfused score + context kernels), full width,
    ragged B = 3, T = 37: forward outputs, losses and every parameter gradient.  fp32: re-association only; bf16: the two walks
    round different intermediates (the python walk builds images of the key GEMM's operands, the library hands the same fp32
    operands to ft_gemm), so the comparison is at operand-rounding level."""
    import flowtron
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, use_cumm_attention=True)
    res = {}
    try:
        for walk in ("python", "fused"):
            os.environ["FLOWTRON_CUMM_LOOP"] = walk
            m, _ = build(cfg, 13, mode)
            b = cuda_batch(synth.make_batch(cfg, [37, 30, 11], [14, 9, 5], seed=13, with_prior=True))
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).sum().backward()
            torch.cuda.synchronize()
            res[walk] = (out[0].detach().cpu(), [a.detach().cpu() for a in out[3]], (nll.item(), gl.item(), ctc.item()),
                         {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    finally:
        os.environ.pop("FLOWTRON_CUMM_LOOP", None)
        os.environ["FLOWTRON_MFMA"] = "f32"
    zp, ap, lp, gp = res["python"]
    zf, af, lf, gf = res["fused"]
    assert mad(zf, zp) < (1e-5 if mode == "f32" else 2e-2)
    for a, b_ in zip(af, ap):
        assert mad(a, b_) < (1e-6 if mode == "f32" else 5e-3)
    for x, y in zip(lf, lp):
        assert abs(x - y) < tol * max(1.0, abs(y))
    worst = ("", 0.0)
    for k, g in gp.items():
        if k.startswith("encoder.convolutions") and k.endswith("conv.bias"):
            continue
        r = (gf[k] - g).norm().item() / max(g.norm().item(), 1e-5 * g.numel() ** 0.5)
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] < (2e-4 if mode == "f32" else 0.1), worst


def test_full_config_invertibility_and_padding_invariance():
    """2-flow default config (config.json): forward(infer(z)) == z (the reference's own, broken,
    test_invertibility bound is 1e-5 'or less', flowtron.py:932-954), and a sample's valid outputs do not
    depend on what it is batched with."""
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    m, _ = build(cfg, 77)
    rs = np.random.RandomState(5)
    N, Lt = 96, 30
    z = torch.from_numpy(rs.standard_normal((1, 80, N)).astype(np.float32)).cuda() * 0.5
    text = torch.from_numpy(rs.randint(0, cfg["n_text"], (1, Lt))).cuda()
    spk = torch.zeros(1, dtype=torch.long).cuda()
    mel, _ = m.infer(z, spk, text, gate_threshold=1.0)
    assert mel.shape == (1, 80, N)
    lens_in, lens_out = torch.tensor([Lt]).cuda(), torch.tensor([N]).cuda()
    out = m(mel, spk, text, lens_in, lens_out)
    zr = out[0].permute(1, 2, 0)
    assert mad(zr, z) < 5e-5, mad(zr, z)
    # same utterance inside a padded batch of 3
    mel3 = torch.zeros(3, 80, N + 17).cuda()
    mel3[0] = torch.from_numpy(rs.standard_normal((80, N + 17)).astype(np.float32)).cuda()
    mel3[1, :, :N] = mel[0]
    mel3[2, :, :40] = mel[0, :, :40]
    text3 = torch.zeros(3, Lt + 6, dtype=torch.long).cuda()
    text3[0] = torch.from_numpy(rs.randint(0, cfg["n_text"], (Lt + 6,))).cuda()
    text3[1, :Lt] = text[0]
    text3[2, :11] = text[0, :11]
    out3 = m(mel3, torch.zeros(3, dtype=torch.long).cuda(), text3, torch.tensor([Lt + 6, Lt, 11]).cuda(),
             torch.tensor([N + 17, N, 40]).cuda())
    assert mad(out3[0][:N, 1], out[0][:, 0]) < 5e-5
    assert mad(out3[1][0][:N, 1], out[1][0][:, 0]) < 5e-5
    assert mad(out3[3][0][1, :N, :Lt], out[3][0][0]) < 1e-5


@pytest.mark.parametrize("mode,width,tol", [("f32", "small", 2e-5), ("bf16", "small", 2e-2), ("bf16", "full", 2e-2)])
def test_batches_wider_than_the_recurrence_kernels_take(mode, width, tol):
    """B = 80 (the reference's nn.LSTM takes any batch, flowtron.py:654-655, :505-512; the launch-per-step kernels and the encoder's
    pair chain stop at 64, one persistent launch at 32): the recurrences run per batch chunk (ops.lstm_layer / bilstm_layer) or, at
    the persistent kernels' geometry, as sliced persistent launches (ft_lstm_persist_*_rows).  The rows of a batch are independent,
    so the first 40 utterances must come out as they do in a batch of their own: z, log_s, attention on valid frames, and every
    gradient of the 80-utterance step equal to the sum of the two 40-utterance halves' (the loss is a sum over utterances once the
    normaliser is taken out)."""
    import flowtron
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG if width == "full" else synth.SMALL_MODEL_CONFIG)
    cfg["n_flows"] = 2
    m, _ = build(cfg, 31, mode)
    try:
        B = 80
        rs = np.random.RandomState(3)
        out_lens = [int(v) for v in rs.randint(20, 61, size=B)]
        in_lens = sorted((int(v) for v in rs.randint(5, 16, size=B)), reverse=True)
        T, Lt = max(out_lens), max(in_lens)

        def sub(lo, hi):                                        # the utterances lo .. hi - 1 as a batch of their own, padded to the same T / L
            b = synth.make_batch(cfg, [T] + out_lens[lo:hi], [Lt] + in_lens[lo:hi], seed=50 + lo, with_prior=True)
            return b

        full = None
        halves = [sub(0, 40), sub(40, 80)]
        # one 80-utterance batch out of the two halves (each half carries a full-length dummy in front that pins T and L: dropped)
        def cat(key, dim=0):
            return torch.cat([h[key][1:] for h in halves], dim)
        full = {k: cat(k) for k in ("mel", "speaker_ids", "text", "in_lens", "out_lens", "gate_target", "attn_prior")}
        order = torch.argsort(full["in_lens"], descending=True, stable=True)      # data.py:200-202: sorted by text length
        assert torch.equal(order, torch.arange(B))

        def run(b):
            for p in m.parameters():
                p.grad = None
            b = cuda_batch(b)
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            n = b["out_lens"].sum().float()
            crit = flowtron.FlowtronLoss(1.0, False, True, False)
            nll, gl, _ = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            ((nll + gl) * n).backward()                          # un-normalised: a sum over utterances
            torch.cuda.synchronize()
            return out, {k: p.grad.detach().clone() for k, p in m.named_parameters()}

        out80, g80 = run(full)
        outs, gs = zip(*[run({k: v[1:] for k, v in h.items()}) for h in halves])
        for hi, (o, lo) in enumerate(zip(outs, (0, 40))):
            for b in range(40):
                t, l = out_lens[lo + b], in_lens[lo + b]
                assert mad(out80[0][:t, lo + b], o[0][:t, b]) <= tol * 10, (hi, b)
                assert mad(out80[3][0][lo + b, :t, :l], o[3][0][b, :t, :l]) <= tol, (hi, b)
        for k in g80:
            ref = gs[0][k] + gs[1][k]
            # (the conv biases in front of an instance norm have analytically zero gradients: |g| ~ 1e-6 of rounding residue, compared
            #  absolutely like in the golden tests -- the recurrences of the wide batch sum in another association, round 6)
            slack = 1e-5 if (k.startswith("encoder.convolutions") and k.endswith("conv.bias")) else 1e-6
            assert float((g80[k] - ref).norm()) <= tol * 5 * float(ref.norm()) + slack, (k, float((g80[k] - ref).norm() / (ref.norm() + 1e-30)))
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"


def test_bf16_mode_close_to_fp32():
    """bf16 MFMA operands, fp32 accumulate/storage: NLL within 2e-2 relative of the fp32 path."""
    import flowtron
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=148, n_flows=2)
    out_lens, in_lens = [120, 96, 64], [30, 24, 18]
    b = cuda_batch(synth.make_batch(cfg, out_lens, in_lens, seed=3, with_prior=True))
    res = {}
    try:
        for mode in ("f32", "bf16"):
            m, _ = build(cfg, 3, mode)
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, _ = flowtron.FlowtronLoss(1.0, False, True, False, 0.0, -8)(out, b["gate_target"], b["in_lens"], b["out_lens"])
            res[mode] = (nll.item(), gl.item(), out[0].detach())
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
    assert abs(res["bf16"][0] - res["f32"][0]) < 2e-2 * abs(res["f32"][0])
    assert abs(res["bf16"][1] - res["f32"][1]) < 5e-2
    assert mad(res["bf16"][2], res["f32"][2]) < 0.25


def test_oracle_agrees_on_fresh_inputs():
    """HIP forward vs the CPU oracle on a new seed (not a stored fixture), 2 flows, ragged batch, no prior."""
    from oracle import flowtron_oracle as O
    from oracle import synth
    cfg = dict(synth.SMALL_MODEL_CONFIG, n_hidden=128, n_attn_channels=64)
    m, sd = build(cfg, 21)
    bc = synth.make_batch(cfg, [37, 30, 12, 5], [14, 9, 9, 3], seed=21, with_prior=False)
    ref = O.forward(sd, cfg, bc["mel"], bc["speaker_ids"], bc["text"], bc["in_lens"], bc["out_lens"], None)
    b = cuda_batch(bc)
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], None)
    assert mad(out[0], ref[0]) < 1e-4
    assert mad(out[2], ref[2]) < 1e-4
    for i in range(2):
        assert mad(out[3][i], ref[3][i]) < 1e-5
        assert mad(out[4][i], ref[4][i]) < 5e-4


def test_stft_mel_vs_reference_golden():
    import audio_processing
    from oracle import synth
    g = _load("stft_mel.pt")
    y = torch.stack([synth.make_audio(g["n_samples"], seed=s) for s in g["seeds"]]).cuda()
    stft = audio_processing.TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0)
    mel = stft.mel_spectrogram(y)
    assert mel.shape == g["mel"].shape
    assert mad(mel, g["mel"]) < 5e-4
    # full 10 s clip: shape contract N//hop + 1 (audio_processing.py:221-225)
    y10 = synth.make_audio(220500, seed=3)[None].cuda()
    assert stft.mel_spectrogram(y10).shape == (1, 80, 862)
    # STFT.transform (audio_processing.py:207-235): magnitude column of the REAL reference's conv1d DFT (fixture mag_b0_f7), and
    # the phase / magnitude pair against torch.stft of the same reflect-padded, hann-windowed signal (|X| abs 2e-3 of ~50, phase
    # where the bin carries energy)
    mag, phase = stft.stft_fn.transform(y)
    assert mag.shape == (2, 513, g["mel"].shape[2]) and phase.shape == mag.shape
    assert mad(mag[0, :, 7], g["mag_b0_f7"]) < 2e-3 * max(1.0, float(g["mag_b0_f7"].abs().max()))
    ref = torch.stft(y.cpu(), 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True), center=True,
                     pad_mode="reflect", return_complex=True)
    assert mad(mag, ref.abs()) < 2e-3 * float(ref.abs().max())
    strong = ref.abs() > 0.05 * ref.abs().max()
    dphi = torch.remainder(phase.cpu() - ref.angle() + math.pi, 2 * math.pi) - math.pi
    assert float(dphi[strong].abs().max()) < 1e-2
    # the general-n_fft kernel (complex radix-2 FFT + dense filterbank) still agrees with the rFFT + sparse-filterbank kernel
    from flowtron_amd import _lib as L
    mel_dense = torch.empty_like(mel)
    st = stft.stft_fn
    L.check(L.lib().ft_stft_mel(L.ptr(y), L.ptr(st.fft_window), L.ptr(stft.mel_basis), L.ptr(mel_dense), 2, y.shape[1], 1024, 256, 80,
                                L.stream()), "ft_stft_mel")
    assert mad(mel, mel_dense) < 2e-4


def test_device_collate_matches_host_collate_and_prior_kernel():
    """flowtron_amd.data.DataCollate(device=cuda): pinned async H2D + the batch prior kernel == host collate + oracle prior."""
    from flowtron_amd.data import DataCollate
    from oracle import flowtron_oracle as O
    torch.manual_seed(3)
    items = [(torch.randn(80, t), torch.tensor([t % 3]), torch.randint(0, 100, (l,))) for t, l in ((37, 9), (52, 14), (20, 14), (45, 3))]
    host = DataCollate(1, False)(items)
    dev = DataCollate(1, True, device="cuda", betab_scaling_factor=1.0)(items)
    for a, b in zip(host[:6], dev[:6]):
        assert b.is_cuda and torch.equal(a, b.cpu())
    pr = dev[6].cpu()
    assert pr.shape == (4, 52, 14)
    for i in range(4):
        L_i, T_i = int(host[3][i]), int(host[4][i])
        ref = O.beta_binomial_prior(L_i, T_i).float()
        assert (pr[i, :T_i, :L_i] - ref).abs().max().item() < 1e-6
        assert float(pr[i, T_i:].abs().max() if T_i < 52 else 0.0) == 0.0


def _oracle_fwd_bwd(cfg, sd, bc, use_ctc=True):
    from oracle import flowtron_oracle as O
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.forward(sdg, cfg, bc["mel"], bc["speaker_ids"], bc["text"], bc["in_lens"], bc["out_lens"], bc["attn_prior"])
    rn, rg, rc = O.loss(ref, bc["gate_target"], bc["in_lens"], bc["out_lens"], 1.0, True, use_ctc, -8)
    (rn + rg + 0.01 * rc).sum().backward()
    return ref, (rn, rg, rc), sdg


def test_batch_of_one_training_matches_oracle():
    """B = 1: the reference takes the UNMASKED instance-norm / unpacked branch (flowtron.py:498, 117-121); forward, the
    three losses and every gradient against the oracle (fp32 MFMA mode)."""
    import flowtron
    from oracle import synth
    cfg = dict(synth.SMALL_MODEL_CONFIG, n_hidden=128, n_attn_channels=64)
    m, sd = build(cfg, 13)
    bc = synth.make_batch(cfg, [27], [10], seed=13, with_prior=True)
    ref, (rn, rg, rc), sdg = _oracle_fwd_bwd(cfg, sd, bc)
    b = cuda_batch(bc)
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    assert mad(out[0], ref[0]) < 1e-4 and mad(out[3][1], ref[3][1]) < 1e-5
    assert abs(nll.item() - rn.item()) < 1e-5 * abs(rn.item()) and abs(gl.item() - rg.item()) < 1e-5
    assert abs(ctc.item() - rc.item()) < 1e-4 * max(1.0, abs(rc.item()))
    for k, p in m.named_parameters():
        r = sdg[k].grad
        assert (p.grad.cpu() - r).norm().item() / max(r.norm().item(), 1e-5 * r.numel() ** 0.5) < 1e-4, k


def test_long_text_many_speakers_ragged_matches_oracle():
    """LibriTTS-shaped edge: 123 speakers, text up to 237 tokens (two 128-column score tiles, 475 CTC states), ragged
    lengths incl. a 1-token / 1-frame-margin sample, T not a multiple of the 32-row attention tile; no prior."""
    import flowtron
    from oracle import synth
    cfg = dict(synth.SMALL_MODEL_CONFIG, n_speakers=123, n_text=185, n_hidden=64, n_attn_channels=48)
    m, sd = build(cfg, 17)
    out_lens, in_lens = [243, 240, 77, 9], [237, 130, 33, 1]
    bc = synth.make_batch(cfg, out_lens, in_lens, seed=17, with_prior=False, n_speakers=123)
    ref, (rn, rg, rc), sdg = _oracle_fwd_bwd(cfg, sd, bc)
    b = cuda_batch(bc)
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], None)
    nll, gl, ctc = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    assert mad(out[0], ref[0]) < 2e-4
    for i in range(2):
        assert mad(out[3][i], ref[3][i]) < 1e-5 and mad(out[4][i], ref[4][i]) < 1e-3
    assert abs(nll.item() - rn.item()) < 1e-5 * abs(rn.item())
    assert abs(ctc.item() - rc.item()) < 2e-4 * max(1.0, abs(rc.item()))
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        r = sdg[k].grad
        e = (p.grad.cpu() - r).norm().item() / max(r.norm().item(), 1e-5 * r.numel() ** 0.5)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 5e-3, worst       # dv = sum de*tanh cancels heavily (rows of de sum to 0); fp32 atomics order its last digits


def test_unsupported_sizes_fail_loudly():
    """error behaviour of the C ABI: a clear message instead of a silent wrong answer or a fallback."""
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    x = torch.randn(4, 70, 16, device="cuda")                       # B = 70 > 64 rows per LSTM step
    w = torch.randn(64, 16, device="cuda")
    with pytest.raises(RuntimeError, match="ft_lstm_seq_fwd"):
        ops.LSTMSeqFn.apply(torch.randn(4, 70, 64, device="cuda"), torch.randn(64, 16, device="cuda"),
                            torch.full((70,), 4, dtype=torch.int32, device="cuda"), False, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(3, 16), w.cpu())
    Q = torch.randn(2, 1, 8, device="cuda")
    with pytest.raises(RuntimeError, match="LDS"):                 # L = 2000 text positions exceed the LDS score tile
        ops.AttentionScoresFn.apply(Q, torch.randn(2000, 1, 8, device="cuda"), torch.randn(1, 8, device="cuda"),
                                    torch.tensor([2000], dtype=torch.int32, device="cuda"), None, 1.0)


def test_bf16_mode_gradients_track_fp32_full_width():
    """Every gradient of the full-width model (H = 1024) in bf16 mode -- two-layer wavefront chain, shared bf16 operand
    images with the dgates hand-off, k-major transpose-read GEMMs, bidirectional pair chain -- against (a) the same model on
    the plain bf16 paths (fp32-staging GEMM, one chain per layer and direction) and (b) the fp32-MFMA run.  A wrong row
    offset / layout flag / stale image shows up as an O(1) error; bf16 operand rounding as ~1e-2.  The encoder conv stack
    is excluded from (b) HERE: its gradients pass three masked instance norms and differ by 0.2-0.4 between bf16 and fp32
    operands on every bf16 path -- including the REAL reference under torch.autocast(bfloat16), which deviates 0.13-0.16 from
    its own fp32 gradients there (tests/golden/cfg2_bf16.pt); tests/test_gpu_bench_path.py holds those groups to that yardstick."""
    import flowtron
    from flowtron_amd import ops
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60)
    b = cuda_batch(synth.make_batch(cfg, [48, 41, 17, 48], [14, 12, 12, 5], seed=6, with_prior=True))
    res = {}
    try:
        for name, mode, new_paths in (("f32", "f32", True), ("bf16", "bf16", True), ("bf16_plain", "bf16", False)):
            ops._BF16_IMAGES = new_paths
            os.environ.update(FLOWTRON_LSTM2="1" if new_paths else "0", FLOWTRON_BILSTM="1" if new_paths else "0")
            m, _ = build(cfg, 6, mode)
            crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).sum().backward()
            torch.cuda.synchronize()
            res[name] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    finally:
        ops._BF16_IMAGES = True
        os.environ.update(FLOWTRON_MFMA="f32", FLOWTRON_LSTM2="1", FLOWTRON_BILSTM="1")

    def worst(a, r, skip=()):
        w = ("", 0.0)
        for k in r:
            if any(k.startswith(p_) for p_ in skip):
                continue
            e = (a[k] - r[k]).norm().item() / max(r[k].norm().item(), 1e-4 * r[k].numel() ** 0.5)
            if e > w[1]:
                w = (k, e)
        return w

    w1 = worst(res["bf16"], res["bf16_plain"])
    assert w1[1] < 2e-2, w1
    w2 = worst(res["bf16"], res["f32"], skip=("encoder.convolutions", "embedding."))
    assert w2[1] < 6e-2, w2


def test_hidden_size_512_model_runs_the_persistent_kernels_and_tracks_fp32():
    """n_hidden = 512 (the reference takes any: flowtron.py:654-655): in the 16-bit modes the attention LSTM and both decoder layers
    run as zero-padded 1024-unit twins on the persistent kernels (ops.lstm_pad_width; AR_Step.forward routes the decoder pair
    through ops.lstm_layer for them).  Loss terms, z and every gradient in bf16 mode against (a) the same model on the
    launch-per-step kernels (ops._PAD_H = False) and (b) the fp32 run -- T 96 so that the padded path is taken."""
    import flowtron
    from flowtron_amd import ops
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60, n_hidden=512)
    b = cuda_batch(synth.make_batch(cfg, [96, 83, 40, 96], [14, 12, 12, 5], seed=8, with_prior=True))
    res = {}
    try:
        for name, mode, pad in (("f32", "f32", True), ("bf16_pad", "bf16", True), ("bf16_step", "bf16", False)):
            ops._PAD_H = pad
            m, _ = build(cfg, 8, mode)
            crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
            if mode == "bf16":
                assert bool(ops.lstm_pad_width(4, 512, False, L_mode(), torch.device("cuda"), 96)) == pad
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).sum().backward()
            torch.cuda.synchronize()
            ops.check_persist_status()
            res[name] = ({k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, float(nll), out[0].detach().cpu())
    finally:
        ops._PAD_H = True
        os.environ.update(FLOWTRON_MFMA="f32")

    def worst(a, r, skip=()):
        w = ("", 0.0)
        for k in r:
            if any(k.startswith(p_) for p_ in skip):
                continue
            e = (a[k] - r[k]).norm().item() / max(r[k].norm().item(), 1e-4 * r[k].numel() ** 0.5)
            if e > w[1]:
                w = (k, e)
        return w

    assert abs(res["bf16_pad"][1] - res["bf16_step"][1]) < 2e-3 * abs(res["bf16_step"][1]) + 1e-4
    assert mad(res["bf16_pad"][2], res["bf16_step"][2]) < 5e-2 and mad(res["bf16_pad"][2], res["bf16_step"][2]) > 0.0
    w1 = worst(res["bf16_pad"][0], res["bf16_step"][0])
    assert w1[1] < 3e-2, w1
    w2 = worst(res["bf16_pad"][0], res["f32"][0], skip=("encoder.convolutions", "embedding."))
    assert w2[1] < 6e-2, w2


def L_mode():
    from flowtron_amd import _lib
    return _lib.mfma_mode()


def test_rccl_process_group_single_rank_step():
    """torch.distributed backend "nccl" (= RCCL) on the GPU box: init through flowtron_amd.dist exactly as bench.py / train.py
    do, the flat-arena wrapper around a small model, one forward/backward, the explicit collectives of the step
    (all_reduce SUM / MAX, broadcast, reduce_tensors) on device tensors.  One rank -- the box has one GPU -- so this pins
    library loading, stream ordering and the hook/callback plumbing, not the wire; the 2-rank numerics are covered on CPU
    with gloo (tests/test_dist_cpu.py).  Runs in a subprocess: a process group is process-global state."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FLOWTRON_MFMA="f32")
        import flowtron
        from flowtron_amd import dist as ftdist
        from flowtron_amd.optim import RAdam
        from oracle import synth
        ftdist.init_distributed(0, 1, "nccl", None)
        assert dist.get_backend() == "nccl"
        cfg = dict(synth.SMALL_MODEL_CONFIG)
        m = flowtron.Flowtron(**cfg)
        m.load_state_dict(synth.make_state_dict(cfg, seed=1))
        m = ftdist.apply_gradient_allreduce(m.cuda().eval())
        opt = RAdam(m.parameters(), lr=1e-3)
        bc = synth.make_batch(cfg, [21, 13], [9, 4], seed=1, with_prior=True)
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in bc.items()}
        crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
        for _ in range(2):
            m.zero_grad()
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).backward()
            opt.clip_grad_norm_(1.0)
            opt.step()
        g = m._grad_arena.flat_grad
        ref = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)              # the step's collective on the real arena (1 rank: identity)
        t = torch.tensor([3.0, 5.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.broadcast(m._grad_arena.flat_param, 0)
        r = ftdist.reduce_tensors([nll, gl, ctc], 1)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(g, ref) and t.tolist() == [3.0, 5.0] and abs(r[0].item() - nll.item()) < 1e-6
        assert torch.isfinite(nll).item()
        dist.destroy_process_group()
        print("RCCL_OK", nll.item())
    ''')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


@pytest.mark.parametrize("out_lens,in_lens", [([3, 2, 1], [2, 1, 1]), ([40] * 33, [9] * 33), ([70, 1], [13, 1])])
def test_bf16_full_width_edge_shapes_match_plain_paths(out_lens, in_lens):
    """Full-width model (H = 1024) in bf16 mode at awkward sizes -- T = 3, a 1-frame / 1-token utterance, B = 33 (three
    16-row MFMA tiles: the both-layer forward kernel does not apply, the skew-2 backward runs with mt = 4) -- new paths
    (image GEMMs, wavefront chain variants, pair chain) against the plain bf16 paths: loss and every gradient."""
    import flowtron
    from flowtron_amd import ops
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=40)
    b = cuda_batch(synth.make_batch(cfg, out_lens, in_lens, seed=9, with_prior=True))
    res = {}
    try:
        for name, new_paths in (("new", True), ("plain", False)):
            ops._BF16_IMAGES = new_paths
            os.environ.update(FLOWTRON_LSTM2="1" if new_paths else "0", FLOWTRON_BILSTM="1" if new_paths else "0")
            m, _ = build(cfg, 9, "bf16")
            crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).sum().backward()
            torch.cuda.synchronize()
            res[name] = (nll.item(), gl.item(), ctc.item(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    finally:
        ops._BF16_IMAGES = True
        os.environ.update(FLOWTRON_MFMA="f32", FLOWTRON_LSTM2="1", FLOWTRON_BILSTM="1")
    for i in range(3):
        assert abs(res["new"][i] - res["plain"][i]) <= 2e-3 * max(1.0, abs(res["plain"][i])), (i, res["new"][i], res["plain"][i])
    worst = ("", 0.0)
    for k, r in res["plain"][3].items():
        e = (res["new"][3][k] - r).norm().item() / max(r.norm().item(), 1e-4 * r.numel() ** 0.5)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 3e-2, worst


def test_three_flows_ragged_bf16_compact_gemms_ignore_stale_memory_in_padded_rows():
    """with n_flows >= 3 the input gradient of the attention LSTM's mel projection reaches the separator rows
    (first padded frame of an utterance) of the EARLIER flows' coupling-output gradients, which their compact weight-gradient GEMMs
    and bias column sums read.  Those rows must be zeros, not whatever the allocator held: 3-flow full-width model, ragged batch,
    the caching allocator's pool poisoned with NaN and with large finite values before every pass; pack-by-length (default)
    against FLOWTRON_GEMM_COMPACT=0 (padded rows multiplied, every padded row written)."""
    import flowtron
    from flowtron_amd import ops
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=40, n_flows=3)
    b = cuda_batch(synth.make_batch(cfg, [37, 30, 22, 37, 9, 3], [11, 10, 9, 8, 4, 2], seed=11, with_prior=True))
    res = {}
    saved = ops._COMPACT
    try:
        for name, compact, poison in (("padded", False, 3.0e4), ("compact_nan", True, float("nan")), ("compact_big", True, 3.0e4)):
            ops._COMPACT = compact
            m, _ = build(cfg, 11, "bf16")
            crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
            junk = [torch.full((n,), poison, device="cuda") for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16)]
            del junk                                                       # back into the allocator's pool, contents intact
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            junk = [torch.full((n,), poison, device="cuda") for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16)]
            del junk
            (nll + gl + 0.01 * ctc).sum().backward()
            torch.cuda.synchronize()
            ops.check_persist_status()
            res[name] = (nll.item(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    finally:
        ops._COMPACT = saved
        os.environ.update(FLOWTRON_MFMA="f32")
    ref = res["padded"]
    for name in ("compact_nan", "compact_big"):
        assert abs(res[name][0] - ref[0]) <= 2e-3 * max(1.0, abs(ref[0]))
        worst = ("", 0.0)
        for k, r in ref[1].items():
            g = res[name][1][k]
            assert torch.isfinite(g).all(), (name, k)
            e = (g - r).norm().item() / max(r.norm().item(), 1e-4 * r.numel() ** 0.5)
            if e > worst[1]:
                worst = (k, e)
        assert worst[1] < 3e-2, (name, worst)


def test_ragged_batch_mel_equals_per_utterance_mel_and_zero_padding():
    """ft_stft_r8_ragged (the data path's collated batch in ONE launch): every utterance's frames are bit-identical
    to TacotronSTFT.mel_spectrogram of that utterance alone (reflect padding about ITS last sample), frames behind an utterance's
    last one are exactly zero (DataCollate's padding, data.py:215-229), including a T_out beyond the longest utterance."""
    from flowtron_amd.audio import TacotronSTFT
    from flowtron_amd.data import DeferredMel
    torch.manual_seed(12)
    stft_args = dict(filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050, mel_fmin=0.0,
                     mel_fmax=8000.0)
    stft = TacotronSTFT(**stft_args).cuda()
    ns = [22050, 9000, 256 * 40 + 255, 256 * 40, 700, 513]
    N = max(ns)
    audio = torch.zeros(len(ns), N)
    for i, n in enumerate(ns):
        audio[i, :n] = (torch.rand(n) * 2 - 1) * 0.8
    n_dev = torch.tensor(ns, dtype=torch.int32, device="cuda")
    T_out = N // 256 + 1 + 3
    mel = stft.mel_spectrogram_ragged(audio.cuda(), n_dev, T_out)
    assert mel.shape == (len(ns), 80, T_out)
    for i, n in enumerate(ns):
        t = n // 256 + 1
        ref = stft.mel_spectrogram(audio[i:i + 1, :n].cuda())[0]
        assert ref.shape[1] == t and torch.equal(mel[i, :, :t], ref), i
        assert float(mel[i, :, t:].abs().max()) == 0.0
    # the loader's deferred slot goes through it
    d = DeferredMel(audio, torch.tensor(ns), stft_args, max_t=T_out).cuda()
    assert torch.equal(d, mel)


@pytest.mark.parametrize("mode,tol", [("f32", 1e-4), ("bf16", 6e-2)])
def test_decoder_lstm_depths_one_and_three_vs_real_reference_golden(mode, tol):
    """VERDICT r3 missing #6: `n_lstm_layers` other than config.json's 2 (flowtron.py:655 passes it to nn.LSTM) -- training runs layer
    after layer through the same projection + recurrence pair; against the REAL reference's z, losses and gradients at depth 1 and 3
    (tests/golden/lstm_depth.pt)."""
    import flowtron
    from oracle import synth
    g = _load("lstm_depth.pt")
    try:
        for c in g["cases"]:
            case, cfg = c["case"], c["cfg"]
            m, _ = build(cfg, case["seed"], mode)
            assert len([k for k in m.state_dict() if k.startswith("flows.0.lstm.weight_ih_l")]) == case["n_lstm_layers"]
            b = cuda_batch(synth.make_batch(cfg, case["out_lens"], case["in_lens"], seed=case["seed"], with_prior=True))
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).backward()
            torch.cuda.synchronize()
            if mode == "f32":
                assert mad(out[0], c["z"]) < 1e-4
                assert abs(nll.item() - c["nll"].item()) < 1e-5 * abs(c["nll"].item())
            else:
                assert abs(nll.item() - c["nll"].item()) < 2e-3 * abs(c["nll"].item())
            worst = ("", 0.0)
            for k, p in m.named_parameters():
                ref = c["grads"][k]
                if mode != "f32" and (k.startswith("encoder.convolutions") or k.startswith("embedding.") or "query" in k):
                    continue                   # ill-conditioned under 16-bit operands for the real reference too (cfg2_bf16.pt)
                if k.startswith("encoder.convolutions") and k.endswith("conv.bias"):
                    continue                   # analytically zero (the instance norm removes the bias): fp32 round-off on both sides
                r = (p.grad.cpu() - ref).norm().item() / max(ref.norm().item(), 1e-5 * ref.numel() ** 0.5)
                if r > worst[1]:
                    worst = (k, r)
            assert worst[1] < tol, (case["n_lstm_layers"], worst)
            # (decode at these depths: test_infer_depth_and_batch_vs_reference_golden; here only that it runs in this operand mode)
            mel, _ = m.infer(torch.randn(1, 80, 4, device="cuda"), b["speaker_ids"][:1], b["text"][:1, :5], gate_threshold=1.0)
            assert mel.shape == (1, 80, 4) and torch.isfinite(mel).all()
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
