"""fp16 operand mode (FT_F16, -m gpu): BASELINE configs[4] is the reference's fp16 AMP run (train.py:211,254,292,323-331 --
`amp.autocast` + `GradScaler`, `scaler.unscale_` + `clip_grad_norm_` + `scaler.step`).  The operand-typed kernel files are
compiled a second time with v_mfma_f32_16x16x32_f16 / v_cvt_pk_f16_f32 (csrc/common.h FT_OPFMT); storage, accumulation and
everything that is not a matmul stay fp32, exactly as in bf16 mode.

Stated tolerances.  fp16 carries 11 significand bits against bf16's 8, so every bf16-mode bound is tightened 4x here
(operand rounding 2^-12 relative vs 2^-9; the remaining factor 2 is head-room):
  GEMM on fp16-rounded operands vs fp64 product of the same rounded operands: 2e-4 abs of an O(1) result (accumulation order only)
  LSTM layer vs fp32 oracle: outputs 8e-3 abs, gradients 8e-3 rel-L2 (bf16: 3e-2)
  full model step (loss scaled by 65536 like GradScaler's initial scale -- unscaled fp16 gradient operands underflow) vs the
  real reference's fp32 gradients (cfg2_bf16.pt): half of what the bf16 path is allowed per parameter
The persistent recurrences must stay BIT-identical to the launch-per-step kernels of the same format.
Range: |x| > 65504 rounds to +-inf (checked), which GradScaler turns into a skipped step (checked)."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
F16 = 2


@pytest.fixture(scope="module")
def env():
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    assert torch.cuda.is_available(), "these tests need the MI355X"
    L.lib()
    return L, ops


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def mad(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


@pytest.mark.parametrize("M,N,K", [(300, 160, 1024), (2048, 1664, 1100), (33, 40, 2000), (65, 7, 130)])
def test_gemm_fp16_operands(env, M, N, K):
    """ft_gemm(FT_F16): the image path (large shapes) and the staging kernel (small ones) on fp16-ROUNDED operands against the
    fp64 product of the same rounded operands."""
    L, ops = env
    torch.manual_seed(M + N)
    A, B = torch.randn(M, K), torch.randn(N, K)
    ref = (A.half().double() @ B.half().double().t()).float() / math.sqrt(K)
    C = torch.empty(M, N, device="cuda")
    ops.gemm_raw(A.cuda(), B.cuda(), C, M, N, K, K, 1, 1, K, N, alpha=1.0 / math.sqrt(K), mode=F16)
    torch.cuda.synchronize()
    assert mad(C, ref) < 2e-4, mad(C, ref)
    # the bf16 build must differ (different rounding of the same inputs): the twin really is another instruction stream
    C2 = torch.empty(M, N, device="cuda")
    ops.gemm_raw(A.cuda(), B.cuda(), C2, M, N, K, K, 1, 1, K, N, alpha=1.0 / math.sqrt(K), mode=1)
    assert mad(C2, ref) > 4 * mad(C, ref)


def test_fp16_saturates_to_inf_beyond_65504(env):
    L, ops = env
    A = torch.full((64, 64), 1.0, device="cuda")
    A[3, 5] = 1.0e5                                       # > 65504: +inf as an fp16 operand
    B = torch.eye(64, device="cuda")
    C = torch.empty(64, 64, device="cuda")
    ops.gemm_raw(A, B, C, 64, 64, 64, 64, 1, 1, 64, 64, mode=F16)
    torch.cuda.synchronize()
    assert torch.isinf(C[3, 5]) and torch.isfinite(C[0]).all()


@pytest.mark.parametrize("T,B,H", [(12, 4, 128), (7, 32, 1024), (9, 20, 256)])
def test_lstm_layer_fp16_vs_fp32_oracle(env, T, B, H):
    """B = 32, H = 1024 runs ft_lstm_persist_{fwd,bwd}_f16; the other shapes the launch-per-step fragment kernels."""
    L, ops = env
    sys.path.insert(0, ROOT)
    from oracle import flowtron_oracle as O
    torch.manual_seed(5 + H)
    I = 24
    lens = torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(T, B, I, requires_grad=True)
    k = 1.0 / math.sqrt(H)
    w_ih, w_hh = [(torch.rand(4 * H, n) * 2 * k - k).requires_grad_(True) for n in (I, H)]
    b_ih, b_hh = [(torch.rand(4 * H) * 2 * k - k).requires_grad_(True) for _ in range(2)]
    O.LSTM_IMPL["fn"] = O.lstm_cell_seq
    ref = O.lstm_cell_seq(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=False)
    go = torch.randn_like(ref)
    ref.backward(go)
    d = [t.detach().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    out = ops.lstm_layer(d[0], lens.int().cuda(), d[1], d[2], d[3], d[4], reverse=False, mode=F16)
    out.backward(go.cuda())
    ops.check_persist_status()
    assert mad(out, ref) < 8e-3, mad(out, ref)
    for mine, r, name in zip(d, (x, w_ih, w_hh, b_ih, b_hh), "x w_ih w_hh b_ih b_hh".split()):
        assert rel(mine.grad, r.grad) < 8e-3, (name, rel(mine.grad, r.grad))


def test_persistent_lstm_fp16_forward_bit_identical_backward_to_rounding(env):
    """the fp16 twins of the persistent recurrences (ft_lstm_roles_fwd_f16: bit-identical to the launch-per-step kernel; the
    reduce-scatter backward ft_lstm_persist_bwd_f16: fp32 rounding, 2e-4 -- 11 significand bits against bf16's 8)"""
    L, _ = env
    from flowtron_amd import ops
    T, B, H = 23, 32, 1024
    if not ops.persist_usable(torch.device("cuda", 0)):
        pytest.skip("persistent kernels not usable on this device")
    lib = L.lib()
    torch.manual_seed(77)
    gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    lens_t = torch.tensor([max(1, T - i) for i in range(B)], dtype=torch.int32, device="cuda")
    status = ops.persist_status(gx.device)
    res = []
    for persist in (False, True):
        y = torch.full((T, B, H), 7.0, device="cuda")
        gates, cell = torch.zeros(T, B, 4 * H, device="cuda"), torch.zeros(T, B, H, device="cuda")
        dgx = torch.full((T, B, 4 * H), 7.0, device="cuda")
        if persist:
            work = torch.empty(lib.ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            ops.roles_launch([ops.fwd_role(gx, lens_t, y, gates, cell, ops.roles_wimg(w, F16, False))], 4, F16, gx.device)
            L.check(lib.ft_lstm_persist_bwd_f16(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work),
                                                L.ptr(status), T, B, H, 21, L.stream()), "bwd")
        else:
            work = torch.empty(lib.ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            L.check(lib.ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                        T, B, H, 0, F16, L.stream()), "fwd")             # FT_F16 dispatches to the twin inside C
            L.check(lib.ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work),
                                        T, B, H, 0, F16, L.stream()), "bwd")
        torch.cuda.synchronize()
        res.append((y, gates, cell, dgx))
    assert ops.check_persist_status()
    act = torch.arange(T, device="cuda")[:, None] < lens_t[None, :]
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][3] - res[1][3]).norm() / res[0][3].norm()) <= 2e-4
    assert torch.equal(res[0][1][act], res[1][1][act]) and torch.equal(res[0][2][act], res[1][2][act])
    # and it is NOT the bf16 result
    y16 = torch.empty(T, B, H, device="cuda")
    work = torch.empty(lib.ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    L.check(lib.ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y16), H, None, None, L.ptr(work), T, B, H, 0, 1, L.stream()), "fwd")
    assert not torch.equal(y16, res[0][0])


def _hip_step(cfg, sd, batch, mode, autocast=None, loss_scale=65536.0):
    """loss_scale: fp16 gradients underflow without it (min normal 6e-5) -- the reference runs fp16 under GradScaler, whose
    initial scale is 65536 (train.py:254); the gradients are unscaled in fp32 afterwards, as scaler.unscale_ does."""
    import flowtron
    os.environ["FLOWTRON_MFMA"] = mode
    try:
        m = flowtron.Flowtron(**cfg)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
        crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
        with torch.autocast("cuda", dtype=autocast, enabled=autocast is not None):
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        ((nll + gl + 0.01 * ctc) * loss_scale).backward()
        torch.cuda.synchronize()
        return (float(nll.detach()), float(gl.detach()), float(ctc.detach())), \
            {k: p.grad.detach().float().cpu() / loss_scale for k, p in m.named_parameters()}
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"


def test_full_width_model_fp16_vs_real_reference_gradients(capsys):
    """cfg2_bf16.pt (the REAL reference, fp32): H = 1024 2-flow model, prior + CTC.  Every parameter's gradient must sit
    within HALF the bf16 path's allowance (bf16: max(0.03, 2 x the reference's own bf16-autocast deviation))."""
    sys.path.insert(0, ROOT)
    from oracle import synth
    g = torch.load(os.path.join(GOLDEN, "cfg2_bf16.pt"), weights_only=False)
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    sd = synth.make_state_dict(cfg, seed=g["seed"])
    batch = synth.make_batch(cfg, g["out_lens"], g["in_lens"], seed=g["seed"], with_prior=True)
    rn, rg, rc = (x.item() for x in g["losses_fp32"])
    results = {}
    for label, mode, ac in (("f16", "f16", None), ("auto+autocast(float16)", "auto", torch.float16)):
        (nll, gl, ctc), grads = _hip_step(cfg, sd, batch, mode, ac)
        assert abs(nll - rn) < 5e-3 * abs(rn) and abs(gl - rg) < 5e-3 and abs(ctc - rc) < 1.5e-2 * abs(rc), (label, nll, rn, gl, rg, ctc, rc)
        rows = []
        for k, e in g["grad"].items():
            mine = grads[k].reshape(-1)
            mine = mine if e["idx"] is None else mine[e["idx"]]
            if k.startswith("encoder.convolutions") and k.endswith("conv.bias"):
                assert mine.abs().max().item() < 1e-4, k
                continue
            dev = rel(mine, e["sample"])
            tol = 0.5 * max(0.03, 2.0 * e["ref_bf16_autocast_rel_dev"])
            rows.append((dev / tol, dev, tol, k))
        rows.sort(reverse=True)
        results[label] = rows
    with capsys.disabled():
        for label, rows in results.items():
            print("\n[%s vs real-reference fp32 gradients] worst:" % label)
            for frac, dev, tol, k in rows[:5]:
                print("   %-58s %.4f (tol %.4f)" % (k, dev, tol))
    for label, rows in results.items():
        assert rows[0][0] < 1.0, (label, rows[0])
    # "auto" under autocast(float16) selects the fp16 kernels (torch ops inside the loss may themselves autocast, so the two
    # runs agree closely rather than bitwise)
    from flowtron_amd import _lib as L
    os.environ["FLOWTRON_MFMA"] = "auto"
    try:
        assert L.mfma_mode() == L.FT_F32
        with torch.autocast("cuda", dtype=torch.float16):
            assert L.mfma_mode() == L.FT_F16
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert L.mfma_mode() == L.FT_BF16
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
    a = {r[3]: r[1] for r in results["f16"]}
    b = {r[3]: r[1] for r in results["auto+autocast(float16)"]}
    assert all(abs(a[k] - b[k]) < 0.1 * a[k] + 1e-4 for k in a)


def test_gradscaler_skips_the_step_on_fp16_overflow_and_steps_otherwise(monkeypatch):
    """train.py:323-331 on the arena views: scaler.scale(loss).backward(); scaler.unscale_(opt); clip_grad_norm_;
    scaler.step(opt); scaler.update().  A scale of 2^40 overflows the fp16 operands of the backward GEMMs -> non-finite
    gradients -> the step is skipped (parameters and RAdam state untouched) and the scale is halved; at 2^10 the step happens."""
    monkeypatch.setenv("FLOWTRON_MFMA", "auto")
    sys.path.insert(0, ROOT)
    import flowtron
    from flowtron_amd.optim import RAdam
    from oracle import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    cfg.update(n_text=64, n_text_dim=128, n_speaker_dim=32, n_attn_channels=64, n_hidden=128)
    sd = synth.make_state_dict(cfg, seed=3)
    batch = synth.make_batch(cfg, [40, 33, 21], [12, 9, 7], seed=3, with_prior=True)
    m = flowtron.Flowtron(**cfg)
    m.load_state_dict(sd)
    m = m.cuda().train()
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}

    def one_step(scaler):
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            loss = nll + gl + 0.01 * ctc
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        scaler.step(opt)
        scaler.update()

    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    s_big = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40)
    torch.manual_seed(5)
    one_step(s_big)
    assert s_big.get_scale() == 2.0 ** 39                           # inf found: backoff
    # (round 5: RAdam sets `_step_supports_amp_scaling`, so GradScaler.step calls step() WITHOUT reading found_inf on the host; the
    # fused kernel drops the update on the non-finite norm and counts it on the device -- the host's count follows at the next look)
    assert all(torch.equal(before[k], p) for k, p in m.named_parameters())
    assert opt.skipped_steps == 1 and opt._step == 0
    s_ok = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    torch.manual_seed(6)
    one_step(s_ok)
    assert s_ok.get_scale() == 2.0 ** 10 and opt._step == 1 and opt.skipped_steps == 1
    assert any(not torch.equal(before[k], p) for k, p in m.named_parameters())
    assert all(torch.isfinite(p).all() for p in m.parameters())
    # the applied update used the schedule of step 1 (calls 2 - drops 1, formed on the device): a fresh optimizer that takes the same
    # step as ITS first one lands on the same parameters (split-K atomics: equal to rounding; step 2's coefficients would be 1.9x off)
    after = {k: p.detach().clone() for k, p in m.named_parameters()}
    m.load_state_dict(sd)
    opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    torch.manual_seed(6)
    one_step(torch.amp.GradScaler("cuda", init_scale=2.0 ** 10))
    for k, p in m.named_parameters():
        d0, d1 = (after[k] - before[k]).float(), (p.detach() - before[k]).float()
        assert float((d0 - d1).abs().max()) <= 2e-2 * float(d0.abs().max()) + 1e-9, k
