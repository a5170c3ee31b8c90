"""Per-kernel parity tests (-m gpu): every C-ABI entry point of the training path is called through
flowtron_amd.ops (ctypes -> libflowtron_hip.so) and compared with a plain fp32 restatement --
the CPU oracle (oracle/flowtron_oracle.py) or a few lines of torch -- on the same seeded inputs.
fp32 MFMA mode (FT_F32, exact fp32 products); tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    assert torch.cuda.is_available(), "these tests need the MI355X"
    L.lib()
    return L, ops


def g(t):
    return t.cuda() if t is not None else None


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def mad(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("mode,tol", [(0, 5e-6), (1, 2e-2)])
@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (37, 45, 19, 1), (300, 160, 1024, 1), (65, 7, 130, 3), (1, 4096, 80, 1), (513, 1, 1664, 1)])
def test_gemm_layouts(env, mode, tol, M, N, K, batch):
    L, ops = env
    torch.manual_seed(M * 7 + N)
    for ta in (False, True):
        for tb in (False, True):
            A = torch.randn(batch, K, M) if ta else torch.randn(batch, M, K)
            Bm = torch.randn(batch, N, K) if tb else torch.randn(batch, K, N)
            bias = torch.randn(N)
            C0 = torch.randn(batch, M, N)
            Am = A.transpose(1, 2) if ta else A
            Bk = Bm.transpose(1, 2) if tb else Bm
            alpha = 1.0 / math.sqrt(K)                 # keeps the pre-activation O(1): bf16 operand rounding ~ 4e-3 abs
            ref = torch.tanh(alpha * (Am.double() @ Bk.double()) + 0.25 * C0.double() + bias.double()).float()
            Ad, Bd, Cd, bd = g(A), g(Bm), g(C0.clone()), g(bias)
            sAm, sAk = (1, M) if ta else (K, 1)
            sBk, sBn = (1, K) if tb else (N, 1)
            ops.gemm_raw(Ad, Bd, Cd, M, N, K, sAm, sAk, sBk, sBn, N, bias=bd, act=L.ACT_TANH, alpha=alpha, beta=0.25,
                         batch=batch, bsA=M * K, bsB=K * N, bsC=M * N, mode=mode)
            torch.cuda.synchronize()
            assert mad(Cd, ref) < tol, (ta, tb, mad(Cd, ref))


def test_gemm_strided_views(env):
    """row-block of a wider weight and a strided output (the no-concat decoder input path)."""
    L, ops = env
    torch.manual_seed(3)
    x = torch.randn(50, 24)
    W = torch.randn(40, 24 + 12)
    y = torch.zeros(50, 40)
    xd, Wd, yd = g(x), g(W), g(y)
    ops.gemm_raw(xd, Wd[:, 12:], yd, 50, 40, 24, 24, 1, 1, 36, 40, mode=0)
    torch.cuda.synchronize()
    assert mad(yd, x @ W[:, 12:].t()) < 1e-5


# ---------------------------------------------------------------- Linear autograd
@pytest.mark.parametrize("act", [0, 1])
def test_linear_two_inputs_autograd(env, act):
    L, ops = env
    torch.manual_seed(11)
    x1 = torch.randn(9, 5, 64, requires_grad=True)
    x2 = torch.randn(9, 5, 48, requires_grad=True)
    W = (torch.randn(33, 112) * 0.1).requires_grad_(True)
    b = torch.randn(33, requires_grad=True)
    pre = torch.cat([x1, x2], 2) @ W.t() + b
    ref = torch.tanh(pre) if act else pre
    go = torch.randn_like(ref)
    ref.backward(go)
    d = [t.detach().cuda().requires_grad_(True) for t in (x1, x2, W, b)]
    out = ops.linear([d[0], d[1]], d[2], d[3], act=act, mode=0)
    out.backward(go.cuda())
    torch.cuda.synchronize()
    assert mad(out, ref) < 1e-5
    for mine, r, name in zip(d, (x1, x2, W, b), "x1 x2 W b".split()):
        assert rel(mine.grad, r.grad) < 1e-5, name


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("K0,K1,N", [(1024, 640, 4096), (64, 48, 256)])
def test_linear_over_two_inputs_as_one_gemm_over_a_concatenated_image(env, monkeypatch, fmt, K0, K1, N):
    """LinearFn over [x0 ; x1] with a row map (the decoder LSTM's input projection, flowtron.py:758-765): ONE compact image of the
    concatenation (ft_bf16_image_rows_into), one K loop over K0 + K1, one split-K weight-gradient GEMM -- against the two-K-piece
    path (FLOWTRON_GEMM_CAT=0: same 16-bit operand rounding, another fp32 summation order) and against fp64 on the valid rows;
    padded rows of y filled with the separator row ("y"), of the input gradients with zeros ("dx")."""
    L, ops = env
    T, B = 23, 7
    torch.manual_seed(K0 + fmt)
    lens_l = [23, 23, 17, 9, 4, 2, 1]
    lens = torch.tensor(lens_l, dtype=torch.int32, device="cuda")
    x0, x1 = torch.randn(T, B, K0, device="cuda"), torch.randn(T, B, K1, device="cuda")
    W, b = torch.randn(N, K0 + K1, device="cuda") * 0.05, torch.randn(N, device="cuda")
    go = torch.randn(T, B, N, device="cuda")
    act = torch.arange(T, device="cuda")[:, None] < lens[None, :]
    go = go * act[..., None]                           # consumers of a compact projection only ever put gradients on valid frames
    res = {}
    for cat in (True, False):
        monkeypatch.setattr(ops, "_CAT_IMAGES", cat)
        d = [t.clone().requires_grad_(True) for t in (x0, x1, W, b)]
        rm = ops.RowMap(lens, T, B)
        y = ops.LinearFn.apply(d[2], d[3], L.ACT_NONE, fmt, rm, "y+dx", d[0], d[1])
        y.backward(go)
        torch.cuda.synchronize()
        res[cat] = (y.detach(), [t.grad.clone() for t in d])
    (ya, ga), (yb, gb) = res[True], res[False]
    tol = 2e-5 if fmt == 2 else 2e-5
    assert float((ya - yb).abs().max()) <= tol * float(yb.abs().max())
    for a_, b_, name in zip(ga, gb, "x0 x1 W b".split()):
        assert float((a_ - b_).norm()) <= 1e-4 * float(b_.norm()) + 1e-12, name
    # fp64 reference with the same operand rounding
    dt = torch.bfloat16 if fmt == 1 else torch.float16
    xr = torch.cat([x0, x1], 2).to(dt).double()
    Wr = W.to(dt).double()
    ref = xr @ Wr.t() + b.double()
    assert float((ya.double() - ref)[act].abs().max()) <= 1e-4 * float(ref.abs().max())
    gor = go.to(dt).double()
    dWr = torch.einsum("tbn,tbk->nk", gor * act[..., None], xr)
    assert float((ga[2].double() - dWr).norm()) <= 2e-4 * float(dWr.norm())
    dxr = (gor @ Wr) * act[..., None]
    assert float((torch.cat([ga[0], ga[1]], 2).double() - dxr).norm()) <= 2e-4 * float(dxr.norm())
    assert float(ga[0][~act].abs().max()) == 0.0 and float(ga[1][~act].abs().max()) == 0.0
    for bi, ln in enumerate(lens_l):                   # "y": padded frames repeat the utterance's first padded frame
        if ln < T - 1:
            assert torch.equal(ya[ln + 1:, bi], ya[ln:ln + 1, bi].expand(T - ln - 1, -1))


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("K0,K1,N", [(1024, 640, 4096), (64, 44, 256)])
def test_gate_layer_on_the_concatenated_image_of_the_decoder_input_projection(env, fmt, K0, K1, N):
    """LinearGateFn (flowtron.py:758-761: the decoder LSTM's input projection and the gate layer both read [h_att ; ctx]): the
    projection as LinearFn's concatenated-image GEMM, the N = 1 gate projection as a GEMV over the SAME image (ft_img_gemv_rows),
    its input gradient as a rank-1 epilogue term of the projection's dX GEMMs, its weight gradient as a GEMV^T -- against
    LinearFn + a separate two-input linear (the path it replaces) and against fp64 with the same operand rounding."""
    L, ops = env
    T, B = 23, 7
    torch.manual_seed(K0 + 7 * fmt)
    lens_l = [23, 23, 17, 9, 4, 2, 1]
    lens = torch.tensor(lens_l, dtype=torch.int32, device="cuda")
    x0, x1 = torch.randn(T, B, K0, device="cuda"), torch.randn(T, B, K1, device="cuda")
    act = torch.arange(T, device="cuda")[:, None] < lens[None, :]
    x0 = x0 * act[..., None]                           # h_att of a padded frame is zero; ctx of all padded frames of b is the same row
    for bi, ln in enumerate(lens_l):
        if ln < T:
            x1[ln:, bi] = x1[ln, bi]
    W, b = torch.randn(N, K0 + K1, device="cuda") * 0.05, torch.randn(N, device="cuda")
    Wg, bg = torch.randn(1, K0 + K1, device="cuda") * 0.1, torch.randn(1, device="cuda")
    go = torch.randn(T, B, N, device="cuda") * act[..., None]
    gg = torch.randn(T, B, 1, device="cuda") * act[..., None]
    rm = ops.RowMap(lens, T, B)
    assert ops.linear_gate_fusable(fmt, rm, [x0, x1], N)
    d = [t.clone().requires_grad_(True) for t in (x0, x1, W, b, Wg, bg)]
    y, gate = ops.LinearGateFn.apply(d[2], d[3], d[4], d[5], fmt, rm, "y+dx", d[0], d[1])
    torch.autograd.backward([y, gate], [go, gg])
    e = [t.clone().requires_grad_(True) for t in (x0, x1, W, b, Wg, bg)]
    y2 = ops.LinearFn.apply(e[2], e[3], L.ACT_NONE, fmt, rm, "y+dx", e[0], e[1])
    gate2 = ops.linear([e[0], e[1]], e[4], e[5], mode=fmt)
    torch.autograd.backward([y2, gate2], [go, gg])
    torch.cuda.synchronize()
    assert torch.equal(y, y2)                          # the same GEMM over the same image
    assert float((gate.detach() - gate2.detach()).abs().max()) <= 2e-5 * float(gate2.detach().abs().max())
    for a_, b_, name in zip(d, e, "x0 x1 W b Wg bg".split()):
        # (the separate linear rounds dgate -- and wg for the input gradients -- to 16-bit GEMM operands; the GEMV^T and the rank-1
        # epilogue term keep them in fp32: the fp64 comparisons below are the yardstick)
        tol = (8e-3 if fmt == 1 else 1e-3) if name == "Wg" else (2e-3 if fmt == 1 else 3e-4) if name in ("x0", "x1") else 1e-4
        assert float((a_.grad - b_.grad).norm()) <= tol * float(b_.grad.norm()) + 1e-12, name
    dt = torch.bfloat16 if fmt == 1 else torch.float16
    xr = torch.cat([x0, x1], 2).to(dt).double()
    ref = xr @ Wg.to(dt).double().t() + bg.double()
    assert float((gate.detach().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())      # every frame: padded ones repeat the separator
    dWg = torch.einsum("tbo,tbk->ok", gg.double(), xr)
    assert float((d[4].grad.double() - dWg).norm()) <= 1e-5 * float(dWg.norm())
    assert abs(float(d[5].grad) - float(gg.double().sum())) <= 1e-5 * float(gg.abs().sum())
    dxr = ((go.to(dt).double() @ W.to(dt).double()) + gg.double() * Wg.double()) * act[..., None]
    assert float((torch.cat([d[0].grad, d[1].grad], 2).double() - dxr).norm()) <= 2e-4 * float(dxr.norm())
    assert float(d[0].grad[~act].abs().max()) == 0.0 and float(d[1].grad[~act].abs().max()) == 0.0


@pytest.mark.parametrize("act_name", ["tanh", "relu", "sigmoid"])
def test_activation_backward_inside_the_image_pass_is_bit_identical_to_the_two_pass_path(env, monkeypatch, act_name):
    """ft_bf16_image_rows_act_bwd (dense layers of the decoder tail, flowtron.py:453-464): dpre = dy act'(pre) formed inside the
    conversion pass (compact image + bias column sums) against ft_act_bwd followed by ft_bf16_image_rows -- the same fp32 products
    rounded once, so the input / weight / bias gradients are bit-identical."""
    L, ops = env
    act = {"tanh": L.ACT_TANH, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID}[act_name]
    T, B, K, N = 29, 6, 256, 512
    torch.manual_seed(3)
    lens = torch.tensor([29, 29, 20, 11, 3, 1], dtype=torch.int32, device="cuda")
    x, W, b = torch.randn(T, B, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda") * 0.1
    valid = (torch.arange(T, device="cuda")[:, None] < lens[None, :])[..., None]
    go = torch.randn(T, B, N, device="cuda") * valid
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(ops, "_FUSE_ACT_BWD", fuse)
        d = [t.clone().requires_grad_(True) for t in (x, W, b)]
        rm = ops.RowMap(lens, T, B)
        y = ops.LinearFn.apply(d[1], d[2], act, L.FT_BF16, rm, "dx", d[0])
        y.backward(go)
        torch.cuda.synchronize()
        res.append([t.grad.clone() for t in d])
    for a_, b_, name in zip(res[0], res[1], "x W b".split()):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_), name
    # and against fp64 on the valid rows
    xr, Wr = x.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
    pre = xr @ Wr.t() + b.double()
    yv = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}[act_name](pre)
    dact = {"tanh": 1 - yv * yv, "relu": (yv > 0).double(), "sigmoid": yv * (1 - yv)}[act_name]
    dpre = (go.double() * dact * valid).to(torch.bfloat16).double()
    dWr = torch.einsum("tbn,tbk->nk", dpre, xr)
    assert float((res[0][1].double() - dWr).norm()) <= 3e-3 * float(dWr.norm())


# ---------------------------------------------------------------- elementwise family
def test_embedding(env):
    L, ops = env
    torch.manual_seed(0)
    W = torch.randn(17, 24, requires_grad=True)
    ids = torch.randint(0, 17, (9, 4))
    ref = W[ids.reshape(-1)]
    go = torch.randn_like(ref)
    ref.backward(go)
    Wd = W.detach().cuda().requires_grad_(True)
    out = ops.embedding(g(ids), Wd)
    out.backward(g(go))
    assert torch.equal(out.cpu(), ref.detach())
    assert mad(Wd.grad, W.grad) < 1e-5
    # ft_embedding_bwd_runs (rows walked at a stride, one atomic per run of equal ids): ids that repeat with the stride (the speaker
    # embedding of a [L,B] batch), arbitrary ids, a row count that is not a multiple of the stride, one chunk and several
    for Lx, Bx, periodic in ((70, 4, True), (70, 4, False), (1, 5, True), (33, 32, True)):
        ids2 = torch.randint(0, 17, (1, Bx)).expand(Lx, Bx).contiguous() if periodic else torch.randint(0, 17, (Lx, Bx))
        ids2 = ids2.reshape(-1)[:Lx * Bx - (1 if Lx > 1 else 0)]
        W2 = torch.randn(17, 24, requires_grad=True)
        go2 = torch.randn(ids2.numel(), 24)
        W2[ids2].backward(go2)
        Wd2 = W2.detach().cuda().requires_grad_(True)
        out2 = ops.embedding(g(ids2), Wd2, run_stride=Bx)
        out2.backward(g(go2))
        assert torch.equal(out2.cpu(), W2.detach()[ids2])
        assert mad(Wd2.grad, W2.grad) < 2e-5, (Lx, Bx, periodic)


def test_reverse_by_length(env):
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(1)
    lens = torch.tensor([9, 4, 1, 7])
    x = torch.randn(9, 4, 5)
    y = ops.reverse_by_length(g(x), g(lens.int()), True)
    assert torch.equal(y.cpu(), O.reverse_by_length(x, lens, 0, 1))
    xb = torch.randn(4, 9, 6)
    yb = ops.reverse_by_length(g(xb), g(lens.int()), False)
    assert torch.equal(yb.cpu(), O.reverse_by_length(xb, lens, 1, 0))
    assert torch.equal(ops.reverse_by_length(yb, g(lens.int()), False).cpu(), xb)       # involution


def test_affine_and_losses(env):
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(2)
    T, B, M = 13, 3, 80
    lens = torch.tensor([13, 9, 4])
    out = (torch.randn(T, B, 2 * M) * 0.3).requires_grad_(True)
    x = torch.randn(T, B, M, requires_grad=True)
    gate = torch.randn(T, B, 1, requires_grad=True)
    target = torch.zeros(B, T)
    for b in range(B):
        target[b, lens[b] - 1:] = 1
    z = torch.exp(out[..., :M]) * x + out[..., M:]
    nll, gl, _ = O.loss((z, [out[..., :M]], gate, None, None), target, None, lens, 1.0, True, False)
    (nll + gl).sum().backward()
    od, xd, gd = (t.detach().cuda().requires_grad_(True) for t in (out, x, gate))
    l32 = g(lens.int())
    zd = ops.AffineFn.apply(od, xd)
    nll_d = ops.NLLFn.apply(zd, l32, 1.0, od[..., :M])
    gl_d = ops.GateBCEFn.apply(gd, g(target), l32)
    (nll_d + gl_d).backward()
    assert mad(zd, z) < 1e-5
    assert abs(nll_d.item() - nll.item()) < 1e-5 * abs(nll.item())
    assert abs(gl_d.item() - gl.item()) < 1e-5
    assert rel(od.grad, out.grad) < 1e-5 and rel(xd.grad, x.grad) < 1e-5 and rel(gd.grad, gate.grad) < 1e-5
    # inverse coupling
    xi = torch.empty_like(zd)
    L.check(L.lib().ft_affine_inv(L.ptr(od.detach()), L.ptr(zd.detach().contiguous()), L.ptr(xi), T * B, M, L.stream()), "inv")
    assert mad(xi, x) < 2e-5


def test_conv_norm_relu(env):
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(4)
    Lx, B, Cc = 11, 3, 32
    lens = torch.tensor([11, 7, 3])
    x = torch.randn(B, Cc, Lx)
    m = O.length_mask(lens, Lx)[:, None, :].float()
    x = (x * m).requires_grad_(True)
    w = (torch.randn(Cc, Cc, 5) * 0.2).requires_grad_(True)
    bconv = torch.randn(Cc, requires_grad=True)
    gam = (1 + 0.1 * torch.randn(Cc)).requires_grad_(True)
    bet = (0.1 * torch.randn(Cc)).requires_grad_(True)
    keep = (torch.rand(B, Cc, Lx) > 0.5).float() * 2
    ref = torch.relu(O.masked_instance_norm(F.conv1d(x * m, w, bconv, padding=2), m, gam, bet)) * keep
    go = torch.randn_like(ref) * m
    ref.backward(go)
    d = [t.detach().cuda().requires_grad_(True) for t in (x.permute(2, 0, 1).contiguous(), w, bconv, gam, bet)]
    out = ops.conv_norm_relu(d[0], g(lens.int()), d[1], d[2], d[3], d[4], g(keep.permute(2, 0, 1).contiguous()), 1e-5, mode=0)
    out.backward(g(go.permute(2, 0, 1).contiguous()))
    refm = (ref * m).permute(2, 0, 1)
    assert mad(out, refm) < 2e-5
    assert rel(d[0].grad, x.grad.permute(2, 0, 1)) < 2e-5
    assert rel(d[1].grad, w.grad) < 2e-5
    assert rel(d[3].grad, gam.grad) < 2e-5 and rel(d[4].grad, bet.grad) < 2e-5
    assert mad(d[2].grad, bconv.grad) < 1e-4          # analytically zero (the norm removes the mean)


# ---------------------------------------------------------------- LSTM
@pytest.mark.parametrize("T,B,I,H,lens", [(9, 3, 20, 64, [9, 5, 2]), (23, 5, 16, 256, [23, 23, 11, 4, 1]),
                                           (6, 33, 12, 32, None), (5, 1, 8, 16, [5])])
@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_seq(env, T, B, I, H, lens, reverse):
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(T + B)
    lens = torch.tensor(lens) if lens is not None else torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(T, B, I, requires_grad=True)
    k = 1.0 / math.sqrt(H)
    w_ih, w_hh = [(torch.rand(4 * H, n) * 2 * k - k).requires_grad_(True) for n in (I, H)]
    b_ih, b_hh = [(torch.rand(4 * H) * 2 * k - k).requires_grad_(True) for _ in range(2)]
    ref = O.lstm_cell_seq(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=reverse)
    go = torch.randn_like(ref)
    ref.backward(go)
    d = [t.detach().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    out = ops.lstm_layer(d[0], g(lens.int()), d[1], d[2], d[3], d[4], reverse=reverse, mode=0)
    out.backward(g(go))
    assert mad(out, ref) < 2e-5, mad(out, ref)
    for mine, r, name in zip(d, (x, w_ih, w_hh, b_ih, b_hh), "x w_ih w_hh b_ih b_hh".split()):
        assert rel(mine.grad, r.grad) < 5e-5, (name, rel(mine.grad, r.grad))


@pytest.mark.parametrize("T,B,H,reverse", [(12, 4, 128, False), (9, 20, 256, True), (7, 32, 1024, False), (5, 33, 512, False),
                                             (6, 3, 96, False)])
def test_lstm_seq_bf16_fragment_path(env, T, B, H, reverse):
    """bf16-operand path (fragment-order W_hh / h / dgates, fused single-launch backward; H % 128 == 0) against the
    fp32 oracle: bf16 operand rounding (2^-9 relative) bounds the error; H = 96 exercises the fp32 fall-back."""
    L, ops = env
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
    ref = O.lstm_cell_seq(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=reverse)
    go = torch.randn_like(ref)
    ref.backward(go)
    d = [t.detach().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    out = ops.lstm_layer(d[0], g(lens.int()), d[1], d[2], d[3], d[4], reverse=reverse, mode=1)
    out.backward(g(go))
    assert mad(out, ref) < 3e-2, mad(out, ref)
    m = O.length_mask(lens, T).t()[..., None]
    assert float((out.cpu() * (~m)).abs().max()) == 0.0                 # pad rows are exactly zero
    for mine, r, name in zip(d, (x, w_ih, w_hh, b_ih, b_hh), "x w_ih w_hh b_ih b_hh".split()):
        assert rel(mine.grad, r.grad) < 3e-2, (name, rel(mine.grad, r.grad))


# ---------------------------------------------------------------- attention
@pytest.mark.parametrize("T,B,Lk,A,E,prior,with_dlp", [(19, 3, 11, 48, 40, True, True), (40, 2, 37, 640, 64, False, True),
                                                        (33, 4, 150, 64, 32, True, False), (5, 1, 3, 20, 8, False, False),
                                                        (300, 2, 20, 64, 16, True, True),       # long T: four 32-row tiles per workgroup, dK combined in LDS
                                                        (70, 3, 23, 100, 16, True, True), (130, 2, 9, 48, 8, False, True)])   # partial a-chunk, idle tile waves
def test_attention(env, T, B, Lk, A, E, prior, with_dlp):
    L, ops = env
    torch.manual_seed(T * 3 + Lk)
    in_lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,))
    in_lens[0] = Lk
    Q = (torch.randn(T, B, A) * 0.7).requires_grad_(True)
    K = (torch.randn(Lk, B, A) * 0.7).requires_grad_(True)
    V = torch.randn(Lk, B, A, requires_grad=True)
    v = (torch.randn(1, A) * 0.3).requires_grad_(True)
    pr = None
    if prior:
        pr = torch.rand(B, T, Lk) ** 3
        pr[0, 0, 0] = 0.0
    temp = 0.9
    pad = ~(torch.arange(Lk)[None, :] < in_lens[:, None])
    e = (torch.tanh(Q.transpose(0, 1)[:, :, None, :] + K.transpose(0, 1)[:, None, :, :]) @ v[0]) / temp
    e = e.masked_fill(pad[:, None, :], -float("inf"))
    p = torch.softmax(e, 2)
    if prior:
        u = torch.log(p + 1e-20) + torch.log(pr + 1e-20)
        lp = u.clone()
        attn = torch.softmax(u.masked_fill(pad[:, None, :], -float("inf")), 2)
    else:
        attn, lp = p, torch.log(p + 1e-8)
    ctx = torch.bmm(attn, V.transpose(0, 1)).transpose(0, 1)
    go = torch.randn_like(ctx)
    valid = (~pad)[:, None, :].float()
    glp = torch.randn_like(lp) * 0.1 * valid
    loss = (ctx * go).sum() + ((lp * glp).sum() if with_dlp else 0.0)
    loss.backward()
    d = [t.detach().cuda().requires_grad_(True) for t in (Q, K, V, v)]
    attn_d, lp_d = ops.AttentionScoresFn.apply(d[0], d[1], d[3], g(in_lens.int()), g(pr), temp)
    ctx_d = ops.ContextFn.apply(attn_d, d[2], 0)
    loss_d = (ctx_d * g(go)).sum() + ((lp_d * g(glp)).sum() if with_dlp else 0.0)
    loss_d.backward()
    assert mad(attn_d, attn) < 2e-6, mad(attn_d, attn)
    assert mad(lp_d, lp) < 2e-4, mad(lp_d, lp)
    assert mad(ctx_d, ctx) < 2e-5
    for mine, r, name in zip(d, (Q, K, V, v), "Q K V v".split()):
        assert rel(mine.grad, r.grad) < 1e-4, (name, rel(mine.grad, r.grad))


# ---------------------------------------------------------------- optimizer kernels
def test_sumsq_radam_colsum(env):
    L, ops = env
    torch.manual_seed(9)
    n = 100003
    p, gr, m, v = torch.randn(n), torch.randn(n) * 3, torch.randn(n) * 0.1, torch.rand(n) * 0.01
    pd, gd, md, vd = g(p), g(gr), g(m), g(v)
    acc = torch.zeros(1, device="cuda")
    part = torch.empty(L.SUMSQ_PARTIALS, device="cuda")
    L.check(L.lib().ft_sumsq(L.ptr(gd), L.ptr(acc), n, L.ptr(part), L.stream()), "sumsq")
    assert abs(acc.item() - (gr.double() ** 2).sum().item()) < 1e-4 * acc.item()
    big = torch.randn(60_977_601, device="cuda") * 1e-3                 # the arena's size (+1: the scalar tail): many workgroups
    seen = set()
    for _ in range(5):                                                  # no float atomics: the same bits every time
        a2 = torch.zeros(1, device="cuda")
        L.check(L.lib().ft_sumsq(L.ptr(big), L.ptr(a2), big.numel(), L.ptr(part), L.stream()), "sumsq")
        seen.add(a2.item())
    assert len(seen) == 1 and abs(seen.pop() - float((big.double() ** 2).sum())) < 1e-4 * float((big.double() ** 2).sum())
    clip, lr, b1, b2, eps, wd, step_size = 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-6, 2.5e-3
    L.check(L.lib().ft_radam_step(L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(vd), n, L.ptr(acc), clip, lr, b1, b2, eps, wd,
                                  step_size, 1, None, L.stream()), "radam")
    cs = min(1.0, clip / (gr.norm().item() + 1e-6))
    g2 = gr * cs
    v2 = v * b2 + (1 - b2) * g2 * g2
    m2 = m * b1 + (1 - b1) * g2
    p2 = p - wd * lr * p
    p2 = p2 - step_size * m2 / (v2.sqrt() + eps)
    assert mad(pd, p2) < 1e-6 and mad(md, m2) < 1e-6 and mad(vd, v2) < 1e-7
    x = torch.randn(1000, 37)
    cs_ = ops.colsum(g(x), 1000, 37, 37)
    assert mad(cs_, x.sum(0)) < 1e-3


def test_beta_binomial_prior_vs_scipy_golden(env):
    """data.py:31-41: golden values produced with scipy.stats.betabinom (tests/golden/make_golden.py)."""
    import os
    L, ops = env
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prior.pt"), weights_only=False)
    pr = ops.beta_binomial_prior(torch.tensor([13, 148]).cuda(), torch.tensor([40, 800]).cuda(), 800, 148)
    assert pr.shape == (2, 800, 148)
    assert mad(pr[0, :40, :13], gold["p13_m40"].float()) < 1e-7
    assert float(pr[0, 40:].abs().max()) == 0.0 and float(pr[0, :, 13:].abs().max()) == 0.0
    ref = gold["p148_m800_s50"].float()
    assert ((pr[1, ::50].cpu() - ref).abs() / (ref + 1e-30)).max().item() < 1e-5       # relative: spans 170 decades


@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_gemm_split_k_weight_gradient_shape(env, beta):
    """few output tiles + long reduction (dW = dpre^T x over T*B rows) takes the split-K / atomic path."""
    L, ops = env
    torch.manual_seed(12)
    rows, N, K = 6000, 130, 200                       # C [N,K] = A^T B, reduction over rows
    dpre, x = torch.randn(rows, N), torch.randn(rows, K)
    bias = torch.randn(K)
    C0 = torch.randn(N, K + 7)                         # strided output (row-block of a wider weight gradient)
    ref = (dpre.double().t() @ x.double()).float() / 64 + bias + beta * C0[:, 3:3 + K]
    Cd = g(C0.clone())
    ops.gemm_raw(g(dpre), g(x), Cd[:, 3:], N, K, rows, 1, N, K, 1, K + 7, bias=g(bias), alpha=1.0 / 64, beta=beta, mode=0, splitk=True)
    torch.cuda.synchronize()
    assert mad(Cd[:, 3:3 + K], ref) < 2e-4
    assert torch.equal(Cd[:, :3].cpu(), C0[:, :3]) and torch.equal(Cd[:, 3 + K:].cpu(), C0[:, 3 + K:])   # neighbours untouched


@pytest.mark.parametrize("blank", [-8.0, -1.0])
def test_attention_ctc_kernel_vs_oracle(env, blank):
    """ft_attn_ctc_fwd/bwd against the oracle's per-sample loop (flowtron.py:162-182 restated, pinned by the golden CTC values)."""
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(21)
    B, T, Lk = 5, 41, 13
    in_lens = torch.tensor([13, 9, 9, 4, 1])
    out_lens = torch.tensor([41, 30, 9, 25, 6])                    # sample 2: T == K (single feasible path)
    lp = torch.log_softmax(torch.randn(B, T, Lk) * 2, 2).requires_grad_(True)
    ref = O.attention_ctc_loss(lp, in_lens, out_lens, blank_logprob=blank)
    (ref * 0.37).backward()
    lpd = lp.detach().cuda().requires_grad_(True)
    mine = ops.AttnCTCFn.apply(lpd, g(in_lens.int()), g(out_lens.int()), blank)
    (mine * 0.37).backward()
    assert abs(mine.item() - ref.item()) < 2e-5 * abs(ref.item()), (mine.item(), ref.item())
    assert mad(lpd.grad, lp.grad) < 2e-6, mad(lpd.grad, lp.grad)
    # infeasible sample (more labels than frames): zero_infinity -> contributes 0 loss and 0 gradient
    out2 = torch.tensor([41, 30, 5, 25, 6])
    lp2 = lp.detach().clone().requires_grad_(True)
    ref2 = O.attention_ctc_loss(lp2, in_lens, out2, blank_logprob=blank)
    ref2.backward()
    lpd2 = lp.detach().cuda().requires_grad_(True)
    mine2 = ops.AttnCTCFn.apply(lpd2, g(in_lens.int()), g(out2.int()), blank)
    mine2.backward()
    assert abs(mine2.item() - ref2.item()) < 2e-5 * abs(ref2.item())
    assert mad(lpd2.grad, torch.nan_to_num(lp2.grad)) < 2e-6
    assert float(lpd2.grad[2].abs().max()) == 0.0


@pytest.mark.parametrize("with_rowmap", [False, True])
def test_input_gradient_of_a_shared_activation_accumulates_in_place(env, monkeypatch, with_rowmap):
    """An activation read by several image-path Linears (h_att: query projection + decoder input projection, flowtron.py:735-765): the
    second consumer's dX GEMM runs with beta = 1 into the first one's buffer and returns None (ops._dx_buffer) instead of leaving a
    [T,B,K] fp32 add to autograd.  Same gradients as the plain path (fp32 rounding of one addition order), pad rows zero where asked."""
    L, ops = env
    torch.manual_seed(77)
    T, B, K = 40, 8, 256
    lens = torch.tensor([40, 33, 33, 20, 11, 9, 4, 1], dtype=torch.int32, device="cuda")
    x0 = torch.randn(T, B, K, device="cuda")
    Ws = [torch.randn(n, K, device="cuda") / K ** 0.5 for n in (128, 64, 96)]
    gs = [torch.randn(T, B, n, device="cuda") for n in (128, 64, 96)]
    res = []
    for inplace in (True, False):
        monkeypatch.setattr(ops, "_DX_INPLACE", inplace)
        x = x0.clone().requires_grad_(True)
        h = ops.AddFn.apply(x, torch.zeros_like(x))                   # a non-leaf activation with three consumers
        rm = ops.row_map(lens, T, B) if with_rowmap else None
        ys = [ops.linear(h, W.clone().requires_grad_(True), None, mode=1, rowmap=rm, fill="y+dx") for W in Ws]
        torch.autograd.backward(ys, gs)
        torch.cuda.synchronize()
        res.append(x.grad.clone())
    assert rel(res[0], res[1]) < 1e-6, rel(res[0], res[1])
    ref = sum((g.reshape(-1, g.shape[-1]).to(torch.bfloat16).double() @ W.to(torch.bfloat16).double()).reshape(T, B, K) for g, W in zip(gs, Ws))
    act = (torch.arange(T, device="cuda")[:, None] < lens[None, :])
    if with_rowmap:
        beyond = (torch.arange(T, device="cuda")[:, None] > lens[None, :])      # (row t == len_b is the utterance's separator: it keeps its value)
        assert float(res[0][beyond].abs().max()) == 0.0 and float(res[1][beyond].abs().max()) == 0.0
        assert rel(res[0][act], ref[act].float()) < 2e-3
    else:
        assert rel(res[0], ref.float()) < 2e-3


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("M,N,K", [(5024, 512, 7680), (300, 128, 4096), (96, 36, 2560)])
def test_deterministic_split_k_is_a_function_of_its_operands(env, fmt, M, N, K):
    """ft_gemm_img with FT_GEMM_SPLITK_DET (the encoder convolutions' forward split-image product): the k-slices' partial products go to a
    workspace and are added in a fixed order -- two runs are BIT-identical (the fp32 atomics of FT_GEMM_SPLITK are not: their order
    varies), the result equals the fp64 product of the rounded operands to fp32 accumulation error, and the un-split GEMM to rounding."""
    L, ops = env
    torch.manual_seed(M + N)
    dt = torch.bfloat16 if fmt == 1 else torch.float16
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    xi, wi = ops.Bf16Image(x, mode=fmt), ops.Bf16Image(w, mode=fmt)
    ref = (x.to(dt).double() @ w.to(dt).double().t() + bias.double()).float()
    outs = []
    for split in ("det", "det", False):
        y = torch.full((M, N), 7.0, device="cuda")
        ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y, M, N, K, N, bias=bias, splitk=split)
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert mad(outs[0], ref) < 2e-4 and mad(outs[2], ref) < 2e-4
    assert L.lib().ft_gemm_img_split_work_bytes(M, N, K) > 0 and L.lib().ft_gemm_img_split_work_bytes(M, N, 1024) == 0   # (short K: one slice)


@pytest.mark.parametrize("n_flows", [1, 2, 3])
def test_fused_flowtron_loss_vs_oracle_and_per_term_path(env, n_flows):
    """ops.FlowtronLossFn (ft_flowtron_loss_fwd/bwd + ft_attn_ctc_fwd/bwd_multi: one autograd node, the odd flows' log-probabilities
    read in reversed time by the kernels) against (a) the oracle's FlowtronLoss restatement (flowtron.py:200-274, which flips and rolls
    them) and (b) the per-term path (NLLFn + GateBCEFn + reverse_by_length + cat + AttnCTCFn) on the same device inputs."""
    L, ops = env
    from oracle import flowtron_oracle as O
    from flowtron_amd.model import FlowtronLoss
    torch.manual_seed(40 + n_flows)
    T, B, M, Lk = 29, 4, 80, 11
    out_lens = torch.tensor([29, 21, 12, 5])
    in_lens = torch.tensor([11, 9, 6, 2])
    outs = [(torch.randn(T, B, 2 * M) * 0.3).requires_grad_(True) for _ in range(n_flows)]
    z = torch.randn(T, B, M, requires_grad=True)
    gate = torch.randn(T, B, 1, requires_grad=True)
    target = torch.zeros(B, T)
    for b in range(B):
        target[b, out_lens[b] - 1:] = 1
    lps = [torch.log_softmax(torch.randn(B, T, Lk) * 2, 2).requires_grad_(True) for _ in range(n_flows)]
    w = (0.7, 1.3, 0.11)

    def total(l3):
        return w[0] * l3[0] + w[1] * l3[1].sum() + w[2] * l3[2].sum()

    ref = O.loss((z, [o[..., :M] for o in outs], gate, [None] * n_flows, lps), target, in_lens, out_lens, 0.8, True, True, blank_logprob=-4.0)
    total(ref).backward()
    crit = FlowtronLoss(sigma=0.8, gate_loss=True, use_ctc_loss=True, ctc_loss_weight=0.1, blank_logprob=-4.0)

    def run(fused):
        old = ops.FUSED_LOSS
        ops.FUSED_LOSS = fused
        try:
            od = [o.detach().cuda().requires_grad_(True) for o in outs]
            zd, gd = z.detach().cuda().requires_grad_(True), gate.detach().cuda().requires_grad_(True)
            ld = [lp.detach().cuda().requires_grad_(True) for lp in lps]
            l3 = crit((zd, [o[..., :M] for o in od], gd, [None] * n_flows, ld), g(target), g(in_lens), g(out_lens))
            total(l3).backward()
            torch.cuda.synchronize()
            return l3, od, zd, gd, ld
        finally:
            ops.FUSED_LOSS = old

    for fused in (True, False):
        l3, od, zd, gd, ld = run(fused)
        assert l3[0].dim() == 0 and l3[1].dim() == 0 and l3[2].dim() == 0
        for a, b in zip(l3, ref):
            assert abs(float(a) - float(b)) < 2e-5 * max(1.0, abs(float(b))), (fused, float(a), float(b))
        assert rel(zd.grad, z.grad) < 1e-5 and rel(gd.grad, gate.grad) < 1e-5
        for a, b in zip(od, outs):
            assert rel(a.grad, b.grad) < 1e-5
            assert float(a.grad[..., M:].abs().max()) == 0.0                       # (the bias half of the coupling output: no loss term)
        for a, b in zip(ld, lps):
            assert mad(a.grad, torch.nan_to_num(b.grad)) < 2e-6, (fused, mad(a.grad, b.grad))
    # gate and CTC switched off: shapes of the reference's placeholders (flowtron.py:237, :245)
    crit0 = FlowtronLoss(sigma=1.0, gate_loss=False, use_ctc_loss=False)
    l3 = crit0((g(z.detach()), [g(outs[0].detach())[..., :M]], None, [None], [None]), g(target), g(in_lens), g(out_lens))
    assert l3[1].shape == (1,) and l3[2].shape == (1,) and float(l3[1]) == 0.0 and float(l3[2]) == 0.0
    ref0 = O.loss((z.detach(), [outs[0].detach()[..., :M]], None, [None], [None]), target, in_lens, out_lens, 1.0, False, False)
    assert abs(float(l3[0]) - float(ref0[0])) < 2e-5 * abs(float(ref0[0]))


@pytest.mark.parametrize("T,B,H", [(9, 20, 256), (7, 32, 1024), (6, 33, 128), (11, 3, 128)])
def test_lstm2_wavefront_chain_vs_oracle_and_unfused(env, T, B, H):
    """csrc/lstm2.hip: both decoder layers as one launch chain (layer 1 one step behind layer 0, input projection of
    layer 1 fused as a second fragment stream) against (a) the fp32 oracle, bf16-operand tolerance, and (b) the unfused
    bf16 path (two ft_lstm_seq_* sequences + a batched GEMM), which differs only by accumulation order."""
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(31 + H)
    I = 40
    lens = torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(T, B, I, requires_grad=True)
    k = 1.0 / math.sqrt(H)
    mk = lambda *shape: (torch.rand(*shape) * 2 * k - k).requires_grad_(True)
    w_ih0, w_hh0, b_ih0, b_hh0 = mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)
    w_ih1, w_hh1, b_ih1, b_hh1 = mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H)
    cpu = [x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1]
    h0 = O.lstm_cell_seq(x, lens, w_ih0, w_hh0, b_ih0, b_hh0)
    ref = O.lstm_cell_seq(h0, lens, w_ih1, w_hh1, b_ih1, b_hh1)
    go = torch.randn_like(ref)
    ref.backward(go)
    l32 = g(lens.int())

    def run(fused):
        d = [t.detach().cuda().requires_grad_(True) for t in cpu]
        if fused:
            gx0 = ops.LinearFn.apply(d[1], d[3] + d[4], L.ACT_NONE, 1, None, "", d[0])
            y = ops.LSTM2SeqFn.apply(gx0, d[2], d[5], d[7], d[8], d[6], l32)
        else:
            y0 = ops.lstm_layer(d[0], l32, d[1], d[2], d[3], d[4], mode=1)
            y = ops.lstm_layer(y0, l32, d[5], d[6], d[7], d[8], mode=1)
        y.backward(g(go))
        return y, d

    yf, df = run(True)
    yu, du = run(False)
    assert mad(yf, ref) < 3e-2 and mad(yu, ref) < 3e-2
    assert mad(yf, yu) < 5e-3, mad(yf, yu)
    m = O.length_mask(lens, T).t()[..., None]
    assert float((yf.detach().cpu() * (~m)).abs().max()) == 0.0
    names = "x w_ih0 w_hh0 b_ih0 b_hh0 w_ih1 w_hh1 b_ih1 b_hh1".split()
    for a, u, r, name in zip(df, du, cpu, names):
        assert rel(a.grad, r.grad) < 4e-2, (name, "fused vs oracle", rel(a.grad, r.grad))
        assert rel(a.grad, u.grad) < 2e-2, (name, "fused vs unfused", rel(a.grad, u.grad))


def test_lstm2_batch_64_four_tiles(env):
    """B = 64 (four 16-row MFMA tiles, the C ABI maximum) through the two-layer wavefront chain."""
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(77)
    T, B, H, I = 5, 64, 128, 24
    lens = torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(T, B, I)
    k = 1.0 / math.sqrt(H)
    mk = lambda *shape: torch.rand(*shape) * 2 * k - k
    w = [mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H), mk(4 * H, H), mk(4 * H, H), mk(4 * H), mk(4 * H)]
    ref = O.lstm_cell_seq(O.lstm_cell_seq(x, lens, w[0], w[1], w[2], w[3]), lens, w[4], w[5], w[6], w[7])
    d = [g(t) for t in w]
    gx0 = ops.linear(g(x), d[0], d[2] + d[3], mode=1)
    y = ops.LSTM2SeqFn.apply(gx0, d[1], d[4], d[6], d[7], d[5], g(lens.int()))
    assert mad(y, ref) < 3e-2


def _bf16_round(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(300, 160, 1024), (1000, 257, 333), (129, 4096, 80), (2048, 1664, 1100), (33, 40, 2000)])
def test_gemm_bf16_image_path(env, M, N, K):
    """gemm_bf16.hip: operand images (k-contiguous / transposing / generic-stride pre-pass) + the DMA-staged NT kernel,
    against an fp64 product of the bf16-ROUNDED operands -- the only error left is fp32 accumulation order (tol 2e-4 of
    an O(1) result), so a wrong swizzle, a missed pad or a stale LDS stage cannot hide inside bf16 rounding noise."""
    L, ops = env
    assert ops._BF16_IMAGES
    torch.manual_seed(M + N + K)
    for ta in (False, True):
        for tb in (False, True):
            A = torch.randn(K, M) if ta else torch.randn(M, K)
            Bm = torch.randn(N, K) if tb else torch.randn(K, N)
            bias, C0 = torch.randn(N), torch.randn(M, N)
            Am = A.t() if ta else A
            Bk = Bm.t() if tb else Bm
            alpha = 1.0 / math.sqrt(K)
            ref = (alpha * (_bf16_round(Am) @ _bf16_round(Bk)) + 0.5 * C0.double() + bias.double()).float()
            Cd = g(C0.clone())
            sAm, sAk = (1, M) if ta else (K, 1)
            sBk, sBn = (1, K) if tb else (N, 1)
            a_args = (g(A), g(Bm), Cd, M, N, K, sAm, sAk, sBk, sBn, N)
            ops.gemm_raw(*a_args, bias=g(bias), alpha=alpha, beta=0.5, mode=1)
            torch.cuda.synchronize()
            assert mad(Cd, ref) < 2e-4, (ta, tb, mad(Cd, ref))
    # generic strides (every 2nd column of a wider matrix) and a source that is only 4-byte aligned
    Aw, Bw = torch.randn(M, 2 * K + 1), torch.randn(N * K + 1)
    Av, Bv = Aw[:, 1::2][:, :K], Bw[1:].view(N, K)
    ref = (_bf16_round(Av) @ _bf16_round(Bv).t()).float() / math.sqrt(K)
    Awd, Bwd = g(Aw), g(Bw)
    Cd = torch.empty(M, N, device="cuda")
    ops.gemm_raw(Awd[:, 1:], Bwd[1:], Cd, M, N, K, 2 * K + 1, 2, 1, K, N, alpha=1.0 / math.sqrt(K), mode=1)
    torch.cuda.synchronize()
    assert mad(Cd, ref) < 2e-4


def test_gemm_bf16_image_path_split_k_and_fallback_agree(env):
    """weight-gradient shape through the image path with atomic split-K; and the staging kernel (no workspace) gives
    the same numbers up to summation order."""
    L, ops = env
    torch.manual_seed(5)
    rows, N, K = 9000, 256, 384
    dpre, x = torch.randn(rows, N), torch.randn(rows, K)
    ref = (_bf16_round(dpre).t() @ _bf16_round(x)).float() / 64
    C1, C2 = torch.empty(N, K, device="cuda"), torch.empty(N, K, device="cuda")
    dd, xd = g(dpre), g(x)
    ops.gemm_raw(dd, xd, C1, N, K, rows, 1, N, K, 1, K, alpha=1.0 / 64, mode=1, splitk=True)
    ops._BF16_IMAGES = False
    try:
        ops.gemm_raw(dd, xd, C2, N, K, rows, 1, N, K, 1, K, alpha=1.0 / 64, mode=1, splitk=True)
    finally:
        ops._BF16_IMAGES = True
    torch.cuda.synchronize()
    assert mad(C1, ref) < 3e-4 and mad(C2, ref) < 3e-4


@pytest.mark.parametrize("B,T", [(5, 7), (32, 19), (40, 3)])
def test_bidirectional_pair_chain_equals_two_chains(env, monkeypatch, B, T):
    """ft_lstm_bidir_seq_fwd/bwd (both directions as the two z-slices of one launch per step) must reproduce the two
    independent ft_lstm_seq_* chains bit for bit -- same kernel bodies, same operands -- outputs and every gradient;
    and the pair against the oracle's bidirectional LSTM (bf16 tolerance)."""
    L, ops = env
    from oracle import flowtron_oracle as O
    torch.manual_seed(B * 100 + T)
    H, I = 128, 48
    lens = torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(T, B, I)
    k = 1.0 / math.sqrt(H)
    mk = lambda *shape: torch.rand(*shape) * 2 * k - k
    ws = [[mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)] for _ in range(2)]
    go = torch.randn(T, B, 2 * H)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FLOWTRON_BILSTM", flag)
        xd = g(x).requires_grad_(True)
        wd = [[g(t).requires_grad_(True) for t in w] for w in ws]
        y = ops.bilstm_layer(xd, g(lens.int()), tuple(wd[0]), tuple(wd[1]), mode=1)
        y.backward(g(go))
        torch.cuda.synchronize()
        res.append([y.detach(), xd.grad] + [t.grad for w in wd for t in w])
    names = ["y", "dx"] + [f"{n}_{d}" for d in "fr" for n in ("dw_ih", "dw_hh", "db_ih", "db_hh")]
    for name, a, b in zip(names, *res):
        if name.startswith("db"):               # ft_colsum combines row blocks with fp32 atomics: order-dependent last bit
            assert mad(a, b) < 1e-5, (name, mad(a, b))
        else:
            assert torch.equal(a, b), (name, mad(a, b))
    ref = torch.cat([O.lstm_cell_seq(x, lens, *ws[0]), O.lstm_cell_seq(x, lens, *ws[1], reverse=True)], 2)
    assert mad(res[0][0], ref) < 3e-2


@pytest.mark.parametrize("fmt", [1, 2])               # FT_BF16 | FT_F16 (the _f16 twins)
@pytest.mark.parametrize("rows,cols,off", [(1000, 300, 0), (257, 4096, 0), (31, 80, 1), (2050, 136, 0)])
def test_bf16_image_and_fused_column_sums(env, rows, cols, off, fmt):
    """ft_bf16_image / ft_bf16_image_colsum: the image is bit-identical to torch's RNE bf16 (fp16) cast, zero outside the
    logical extent ([ceil256(rows+32)][ceil256(cols)]); the fused column sums equal the fp32 sums of the SOURCE
    (tolerance: fp32 summation order over <= 2050 rows)."""
    L, ops = env
    torch.manual_seed(rows + cols)
    wide = torch.randn(rows, cols + off + 3)
    # exact ties of both formats, fp16 subnormals, the fp16 overflow boundary (65520 is the first value that rounds to inf)
    special = [1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11, 1 + 2.0 ** -8, 1 + 3 * 2.0 ** -8, 3e-6, -3e-6, 6.0e-8, 65519.0, 65520.0, -1e5]
    wide[0, off:off + len(special)] = torch.tensor(special)
    src = g(wide)[:, off:off + cols]                       # strided view; off = 1 makes the rows only 4-byte aligned
    for with_sum in (False, True):
        img = ops.Bf16Image(src, colsum=with_sum, mode=fmt)
        Rp, ld = (rows + 32 + 255) // 256 * 256, (cols + 255) // 256 * 256
        assert img.ld == ld and img.buf.numel() >= Rp * ld * 2
        raw = img.buf[:Rp * ld * 2].view(torch.int16).view(Rp, ld).cpu()
        ref = wide[:, off:off + cols].to(torch.bfloat16 if fmt == 1 else torch.float16).view(torch.int16)
        assert torch.equal(raw[:rows, :cols], ref)
        assert int(raw[rows:].abs().max()) == 0 and (cols == ld or int(raw[:, cols:].abs().max()) == 0)
        if with_sum:
            view = wide[:, off:off + cols]
            err = (img.colsum.cpu() - view.double().sum(0).float()).abs()
            assert bool((err < 2e-4 * math.sqrt(rows) + 1e-6 * view.abs().sum(0)).all()), err.max()


def _bench_lens(lens, ng):
    """lens == "bench": the ragged out_lens of bench.py's own batch (BASELINE configs[1]: B = 32, T = 862), i.e. the exact launch
    geometry the benchmark times; only for the default transport (ng = 1) -- the others are covered at the short shapes."""
    if lens != "bench":
        return lens
    import bench
    return [int(v) for v in bench.synth_batch(32, 1234 + 7)["out_lens"]]


@pytest.mark.parametrize("T,B,lens", [(37, 32, None), (9, 5, [9, 9, 4, 2, 1]), (20, 17, None), (862, 32, "bench")])
def test_persistent_lstm_forward_is_bit_identical_to_launch_per_step(env, T, B, lens):
    """the persistent forward recurrence (ft_lstm_roles_fwd at 4 rows per XCD group, csrc/lstm_roles.hip: one launch per sequence, W_hh
    fragments resident in registers, bare operand pairs behind a sentinel) against ft_lstm_seq_fwd(FT_BF16): same rounding and
    accumulation order -> bit-identical y / saved gates / saved cell on every valid (t, b), zeros on pad rows, status word clean --
    at the short shapes and at the bench's own launch geometry (tests/test_gpu_roles.py: every other R / windowing / role placement)."""
    L, ops = env
    H = 1024
    if not ops.persist_usable(torch.device("cuda", 0)):
        pytest.skip("persistent kernels not usable on this device")
    torch.manual_seed(T * 100 + B)
    gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    lens = _bench_lens(lens, 1)
    if lens is None:
        lens = [max(1, T - 2 * i) for i in range(B)]
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    outs = []
    for persist in (False, True):
        y = torch.full((T, B, H), 7.0, device="cuda")
        gates = torch.zeros(T, B, 4 * H, device="cuda")
        cell = torch.zeros(T, B, H, device="cuda")
        if persist:
            ops.roles_launch([ops.fwd_role(gx, lens_t, y, gates, cell, ops.roles_wimg(w, 1, False))], 4, 1, gx.device)
        else:
            work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                            T, B, H, 0, 1, L.stream()), "ft_lstm_seq_fwd")
        torch.cuda.synchronize()
        outs.append((y, gates, cell))
    assert ops.check_persist_status()
    act = torch.arange(T, device="cuda")[:, None] < lens_t[None, :]
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1][act], outs[1][1][act]) and torch.equal(outs[0][2][act], outs[1][2][act])
    assert float(outs[1][0][~act].abs().max() if (~act).any() else 0.0) == 0.0


@pytest.mark.parametrize("B", [33, 48, 64, 100])
def test_wide_batch_runs_on_wider_xcd_groups(env, B, monkeypatch):
    """B > 32 (the reference's nn.LSTM has no batch limit, flowtron.py:654-655): ops.LSTMSeqFn runs the persistent recurrences with 8
    (B <= 64) or 16 rows per XCD group forward, slices of 64 rows at 8 per group backward (ops.roles_plan).  Forward: bit-identical to
    the launch-per-step kernels (valid rows; zeros on pad rows); backward (reduce-scatter form): equal to fp32 rounding, like the
    single-launch case; W_hh gradient from both."""
    L, ops = env
    H, T = 1024, 21
    if not ops.persist_usable(torch.empty(1, device="cuda").device):
        pytest.skip("persistent kernels not usable on this device")
    torch.manual_seed(900 + B)
    gx0 = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w0 = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    lens = torch.tensor([max(1, T - (i * 7) % T) for i in range(B)], dtype=torch.int32, device="cuda")
    act = (torch.arange(T, device="cuda")[:, None] < lens[None, :])
    res = []
    for wide in (True, False):
        monkeypatch.setenv("FLOWTRON_LSTM_PERSIST", "1" if wide else "0")
        assert bool(ops.lstm_persist_slices(B, H, False, 1, gx0.device)) == wide
        n0 = ops.PERSIST_LAUNCHES
        gx, w = gx0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        if wide or B <= ops.MAX_STEP_BATCH:
            y = ops.LSTMSeqFn.apply(gx, w, lens, False, 1)
        else:                                                   # (the launch-per-step kernels take 64 rows: per batch chunk, like ops.lstm_layer)
            y = torch.cat([ops.LSTMSeqFn.apply(gx[:, b0:b0 + nb].contiguous(), w, lens[b0:b0 + nb].contiguous(), False, 1)
                           for b0, nb in ops.batch_chunks(B, ops.MAX_STEP_BATCH)], 1)
        (y * dy).sum().backward()
        torch.cuda.synchronize()
        assert ops.check_persist_status()
        assert (ops.PERSIST_LAUNCHES - n0 > 0) == wide
        res.append((y.detach(), gx.grad, w.grad))
    (y1, g1, dw1), (y0, g0, dw0) = res
    assert torch.equal(y1, y0)
    assert float(y1[~act].abs().max()) == 0.0 and float(g1[~act].abs().max()) == 0.0
    # (reduce-scatter backward: the same products in another fixed association, rel-L2 ~1e-4 like the single-launch case)
    assert rel(g1, g0) < 5e-4 and rel(dw1, dw0) < 2e-3, (rel(g1, g0), rel(dw1, dw0))


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("ng", [21])
@pytest.mark.parametrize("T,B,lens", [(37, 32, None), (9, 5, [9, 9, 4, 2, 1]), (40, 32, "ragged0"), (862, 32, "bench")])
def test_persistent_backward_emits_the_compact_dgates_image(env, fmt, ng, T, B, lens):
    """ft_lstm_persist_bwd_img: the same dgx as ft_lstm_persist_bwd, plus the compact 16-bit image of dgates -- bit-identical to
    ft_bf16_image_rows over that dgx (valid rows, zero separators, zero rows up to ceil256(R + 32)) -- and its column sums."""
    L, ops = env
    H = 1024
    if not L.lib().ft_lstm_persist_supported(B, H):
        pytest.skip("needs a 256-CU device")
    torch.manual_seed(T * 10 + B + fmt)
    gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    if lens == "ragged0":
        lens = [T] + [max(1, (T * (B - i)) // B - (i % 3)) for i in range(1, B)]
    else:
        lens = _bench_lens(lens, ng)
    if lens is None:
        lens = [max(1, T - 2 * i) for i in range(B)]
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    y, gates, cell = torch.empty(T, B, H, device="cuda"), torch.zeros(T, B, 4 * H, device="cuda"), torch.zeros(T, B, H, device="cuda")
    work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                    T, B, H, 0, fmt, L.stream()), "ft_lstm_seq_fwd")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    d0 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    d1 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    L.check(L.op16("ft_lstm_persist_bwd", fmt)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(d0), L.ptr(wp),
                                               L.ptr(status), T, B, H, ng, L.stream()), "ft_lstm_persist_bwd")
    rm = ops.RowMap(lens_t, T, B)
    img = ops.Bf16Image.empty_rows(4 * H, rm, fmt, torch.device("cuda"))
    img.buf.fill_(0x5A)                                                 # whatever the allocator left behind
    rows_alloc = img.buf.numel() // (2 * img.ld)
    L.check(L.op16("ft_lstm_persist_bwd_img", fmt)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(d1), L.ptr(wp),
                                                   L.ptr(status), T, B, H, ng, L.ptr(img.buf), img.ld, rows_alloc, L.ptr(img.colsum),
                                                   L.stream()), "ft_lstm_persist_bwd_img")
    img2 = ops.Bf16Image.empty_rows(4 * H, rm, fmt, torch.device("cuda"))          # image ONLY: no fp32 dgx at all
    img2.buf.fill_(0x33)
    L.check(L.op16("ft_lstm_persist_bwd_img", fmt)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), None, L.ptr(wp),
                                                   L.ptr(status), T, B, H, ng, L.ptr(img2.buf), img2.ld, rows_alloc, L.ptr(img2.colsum),
                                                   L.stream()), "ft_lstm_persist_bwd_img (image only)")
    ref = ops.Bf16Image(d0.reshape(T * B, 4 * H), colsum=True, mode=fmt, rowmap=rm)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(d0, d1)
    R = sum(lens) + B
    Rz = min((R + 32 + 255) // 256 * 256, rows_alloc)
    a = img.buf[: Rz * img.ld * 2].view(torch.int16).view(Rz, img.ld)[:, : 4 * H]
    b = ref.buf[: Rz * ref.ld * 2].view(torch.int16).view(Rz, ref.ld)[:, : 4 * H]
    assert torch.equal(a, b)
    assert float((img.colsum - ref.colsum).abs().max()) <= 1e-4 * float(ref.colsum.abs().max()) + 1e-6
    a2 = img2.buf[: Rz * img2.ld * 2].view(torch.int16).view(Rz, img2.ld)[:, : 4 * H]
    assert torch.equal(a2, b)
    assert float((img2.colsum - ref.colsum).abs().max()) <= 1e-4 * float(ref.colsum.abs().max()) + 1e-6


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("T,B,lens", [(37, 32, None), (9, 5, [9, 9, 4, 2, 1]), (40, 32, "ragged0"), (1, 32, None), (862, 32, "bench")])
def test_reduce_scatter_backward_recurrence_matches_the_launch_per_step_kernel(env, fmt, T, B, lens):
    """Transport 21 (lstm_persist_bwd_rs_k, the step's default backward recurrence since round 4): every CU multiplies its OWN dgates
    with its 128 rows of W_hh and the fp32 partials are reduce-scattered through the XCD's L2, so the sum over the recurrent product
    runs in another order than lstm_bwd_step_bf16's -- equal to fp32 rounding, not bit for bit.  Held against the launch-per-step
    kernel (same 16-bit operand rounding of dgates): a different fp32 rounding occasionally flips the 16-bit rounding of a dgates
    element (one part in 2^9 / 2^12 of that element), which the contracting recurrence carries along: rel-L2 <= 1e-3 (bf16; observed
    2.5e-4 at T 862) / 2e-4 (fp16), and the first step (no recurrent term) equal to a few fp32 ulps (the cell backward's products are re-associated).  The three output modes agree bit for bit
    among themselves: fp32 rows, rows + compact 16-bit image, image only (= ft_bf16_image_rows of the rows, and the column sums)."""
    L, ops = env
    H = 1024
    if not L.lib().ft_lstm_persist_supported(B, H):
        pytest.skip("needs a 256-CU device")
    torch.manual_seed(T * 7 + B + fmt)
    gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    if lens == "ragged0":
        lens = [T] + [max(1, (T * (B - i)) // B - (i % 3)) for i in range(1, B)]
    else:
        lens = _bench_lens(lens, 1)
    if lens is None:
        lens = [max(1, T - 2 * i) for i in range(B)]
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    y, gates, cell = torch.empty(T, B, H, device="cuda"), torch.zeros(T, B, 4 * H, device="cuda"), torch.zeros(T, B, H, device="cuda")
    work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                    T, B, H, 0, fmt, L.stream()), "ft_lstm_seq_fwd")
    d0 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(d0), L.ptr(work),
                                    T, B, H, 0, fmt, L.stream()), "ft_lstm_seq_bwd")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    d1 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    L.check(L.op16("ft_lstm_persist_bwd", fmt)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(d1), L.ptr(wp),
                                               L.ptr(status), T, B, H, 21, L.stream()), "ft_lstm_persist_bwd (21)")
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    act = torch.arange(T, device="cuda")[:, None] < lens_t[None, :]
    assert float(d1[~act].abs().max() if (~act).any() else 0.0) == 0.0            # padded frames: exact zeros
    rel = float((d0 - d1).norm() / d0.norm())
    assert rel <= (1e-3 if fmt == 1 else 2e-4), rel
    for b in range(B):          # each utterance's LAST frame has no recurrent term: only the re-associated products of the cell backward
        a_, b_ = d0[lens[b] - 1, b], d1[lens[b] - 1, b]
        assert float((a_ - b_).abs().max()) <= 2e-6 * float(a_.abs().max()) + 1e-12, b
    # output modes: rows + image, image only
    rm = ops.RowMap(lens_t, T, B)
    imgs = []
    for with_rows in (True, False):
        img = ops.Bf16Image.empty_rows(4 * H, rm, fmt, torch.device("cuda"))
        img.buf.fill_(0x5A)
        rows_alloc = img.buf.numel() // (2 * img.ld)
        d2 = torch.full((T, B, 4 * H), 7.0, device="cuda")
        L.check(L.op16("ft_lstm_persist_bwd_img", fmt)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell),
                                                       L.ptr(d2) if with_rows else None, L.ptr(wp), L.ptr(status), T, B, H, 21,
                                                       L.ptr(img.buf), img.ld, rows_alloc, L.ptr(img.colsum), L.stream()), "ft_lstm_persist_bwd_img (21)")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        if with_rows:
            assert torch.equal(d1, d2)
        imgs.append(img)
    ref = ops.Bf16Image(d1.reshape(T * B, 4 * H), colsum=True, mode=fmt, rowmap=rm)
    torch.cuda.synchronize()
    R = sum(lens) + B
    Rz = min((R + 32 + 255) // 256 * 256, rows_alloc)
    rb = ref.buf[: Rz * ref.ld * 2].view(torch.int16).view(Rz, ref.ld)[:, : 4 * H]
    for img in imgs:
        a = img.buf[: Rz * img.ld * 2].view(torch.int16).view(Rz, img.ld)[:, : 4 * H]
        assert torch.equal(a, rb)
        assert float((img.colsum - ref.colsum).abs().max()) <= 1e-4 * float(ref.colsum.abs().max()) + 1e-6


def test_image_only_gradient_fails_loudly_when_read_as_fp32(env):
    """ops._require_written: a gradient that exists only as its 16-bit image must never be read as fp32 by a consumer that missed
    the hand-off"""
    L, ops = env
    t = torch.empty(4, 8, device="cuda")
    ops._HANDOFF["unwritten"].add((t.device.index, t.data_ptr(), tuple(t.shape)))
    try:
        with pytest.raises(RuntimeError, match="only as a 16-bit operand image"):
            ops._require_written(t)
        ops._require_written(torch.empty(4, 8, device="cuda"))             # any other tensor: fine
    finally:
        ops._handoff_clear()


def test_image_only_gradient_reads_nan_for_a_foreign_consumer_and_anomaly_mode_gets_values(env, monkeypatch):
    """VERDICT r3 #6 / SURVEY 8(b) "gradients remain ordinary": between the persistent backward recurrence and the input projection's
    backward the default path carries the dgates gradient as its 16-bit image only.  A foreign reader between the two nodes (a
    node pre-hook here) must see NaN in every element -- never stale allocator memory --, the repo's own consumer still produces the
    gradients of the `both` path bit for bit, no gradient-sized fp32 buffer is allocated, and anomaly mode (which inspects every
    returned gradient) is served real values."""
    L, ops = env
    H, T, B, K = 1024, 12, 32, 64
    if not ops.lstm_persist_groups(B, H, False, L.FT_BF16, torch.device("cuda", torch.cuda.current_device())):
        pytest.skip("persistent recurrences not usable on this device")
    torch.manual_seed(5)
    lens = torch.tensor([T] * 4 + [max(2, T - i) for i in range(B - 4)], dtype=torch.int32, device="cuda")
    x0 = torch.randn(T, B, K, device="cuda")
    w_ih0, w_hh0 = torch.randn(4 * H, K, device="cuda") * 0.1, torch.randn(4 * H, H, device="cuda") / H ** 0.5
    b = torch.zeros(4 * H, device="cuda")
    dy = torch.randn(T, B, H, device="cuda") * 0.1

    def run(img_mode, hook, anomaly=False):
        monkeypatch.setattr(ops, "_PERSIST_IMG", img_mode)
        monkeypatch.setattr(ops, "_GX16", False)                          # (bitwise comparison across the modes: gx as fp32 rows in all of them)
        torch.empty(64 << 20, device="cuda").fill_(7.0)                  # leave recognisable garbage in the allocator's pool
        x, w_ih, w_hh = x0.clone().requires_grad_(True), w_ih0.clone().requires_grad_(True), w_hh0.clone().requires_grad_(True)
        rm = ops.RowMap(lens, T, B)
        seen = []
        import contextlib
        with (torch.autograd.detect_anomaly(check_nan=True) if anomaly else contextlib.nullcontext()):
            h = ops.lstm_layer(x, lens, w_ih, w_hh, b, b, mode=L.FT_BF16, rowmap=rm, fill="dx")
            lin_node = h.grad_fn.next_functions[0][0]
            assert "LinearFn" in type(lin_node).__name__
            if hook:
                lin_node.register_prehook(lambda grads: seen.append(grads[0]))
            h.backward(dy)
        torch.cuda.synchronize()
        ops.check_persist_status()
        return x.grad, w_ih.grad, w_hh.grad, seen

    gx_b, gwi_b, gwh_b, _ = run("both", False)
    gx_1, gwi_1, gwh_1, seen = run("1", True)
    assert len(seen) == 1
    g = seen[0]
    assert g.shape == (T, B, 4 * H) and bool(torch.isnan(g).all()), "a foreign reader must see NaN, not unwritten memory"
    assert g.untyped_storage().nbytes() <= 1024, "no gradient-sized fp32 buffer behind an image-only gradient"
    for a_, b_ in ((gx_1, gx_b), (gwi_1, gwi_b), (gwh_1, gwh_b)):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_)
    gx_a, gwi_a, gwh_a, _ = run("1", False, anomaly=True)               # anomaly mode: real values, no "returned nan" error
    for a_, b_ in ((gx_a, gx_b), (gwi_a, gwi_b), (gwh_a, gwh_b)):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("T,B,lens", [(157, 32, "ragged"), (9, 5, [9, 9, 4, 2, 1]), (40, 17, None), (33, 32, "full")])
def test_persistent_bilstm_matches_the_launch_per_step_pair_chain(env, fmt, T, B, lens):
    """csrc/bilstm_persist.hip (the text encoder's BiLSTM, H 256, one launch per pass for both directions) against the pair chain
    ft_lstm_bidir_seq_*: same 16-bit operand rounding, different fp32 summation order (K is not split over waves) -> y, saved gates /
    cell and dgx agree to rounding; y and dgx are exactly zero on padded frames."""
    L, ops = env
    H = 256
    if not L.lib().ft_bilstm_persist_supported(B, H):
        pytest.skip("needs a 256-CU device")
    torch.manual_seed(T + B + fmt)
    if lens == "ragged":
        lens = [T] + [max(1, T - 4 * i - (i % 3)) for i in range(1, B)]
    elif lens == "full":
        lens = [T] * B
    elif lens is None:
        lens = [max(1, T - 2 * i) for i in range(B)]
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    gx = [torch.randn(T, B, 4 * H, device="cuda") * 0.7 for _ in range(2)]
    w = [torch.randn(4 * H, H, device="cuda") / H ** 0.5 for _ in range(2)]
    dy = torch.randn(T, B, 2 * H, device="cuda") * 0.1
    f = dict(device="cuda", dtype=torch.float32)
    res = []
    for persistent in (False, True):
        y = torch.full((T, B, 2 * H), 7.0, **f)
        gates = [torch.zeros(T, B, 4 * H, **f) for _ in range(2)]
        cell = [torch.zeros(T, B, H, **f) for _ in range(2)]
        dgx = [torch.full((T, B, 4 * H), 7.0, **f) for _ in range(2)]
        if persistent:
            status = torch.zeros(1, dtype=torch.int32, device="cuda")
            wk = torch.empty(L.lib().ft_bilstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            L.check(L.op16("ft_bilstm_persist_fwd", fmt)(L.ptr(gx[0]), L.ptr(gx[1]), L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(y), 2 * H,
                                                         L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(wk),
                                                         L.ptr(status), T, B, H, L.stream()), "ft_bilstm_persist_fwd")
            L.check(L.op16("ft_bilstm_persist_bwd", fmt)(L.ptr(dy), 2 * H, L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(gates[0]),
                                                         L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(dgx[0]), L.ptr(dgx[1]),
                                                         L.ptr(wk), L.ptr(status), T, B, H, L.stream()), "ft_bilstm_persist_bwd")
            torch.cuda.synchronize()
            assert int(status.item()) == 0
        else:
            wk = [torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8) for _ in range(2)]
            L.check(L.op16("ft_lstm_bidir_seq_fwd", fmt)(L.ptr(gx[0]), L.ptr(gx[1]), L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(y), 2 * H,
                                                         L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(wk[0]),
                                                         L.ptr(wk[1]), T, B, H, L.stream()), "ft_lstm_bidir_seq_fwd")
            L.check(L.op16("ft_lstm_bidir_seq_bwd", fmt)(L.ptr(dy), 2 * H, L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(gates[0]),
                                                         L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(dgx[0]), L.ptr(dgx[1]),
                                                         L.ptr(wk[0]), L.ptr(wk[1]), T, B, H, L.stream()), "ft_lstm_bidir_seq_bwd")
            torch.cuda.synchronize()
        res.append((y, gates, cell, dgx))
    act = torch.arange(T, device="cuda")[:, None] < lens_t[None, :]
    (y0, g0, c0, d0), (y1, g1, c1, d1) = res
    assert float(y1[~act].abs().max() if (~act).any() else 0.0) == 0.0
    # a 1e-7 difference in a pre-activation can flip the 16-bit rounding of h_t (1 ulp = 4e-3 relative), and 157 steps carry it
    # on: the two implementations sit 3-9e-4 apart and EQUALLY far (1.06e-3) from the fp32 oracle (scripts/exp/bilstm_persist_check.py)
    tol = 3e-3 if fmt == 1 else 6e-4
    assert float((y0 - y1).abs().max()) < tol
    for d in range(2):
        assert float((g0[d][act] - g1[d][act]).abs().max()) < tol and float((c0[d][act] - c1[d][act]).abs().max()) < 2 * tol
        assert float(d1[d][~act].abs().max() if (~act).any() else 0.0) == 0.0
        scale = float(d0[d].abs().max())
        assert float((d0[d] - d1[d]).abs().max()) < tol * scale + 1e-7, (d, float((d0[d] - d1[d]).abs().max()), scale)
    # against the definition (fp32 CPU oracle, explicit recurrence): the bf16 / fp16 operand tolerance of the other LSTM tests
    from oracle import flowtron_oracle as O
    lens_c = torch.tensor(lens)
    for d in range(2):
        g = gx[d].cpu().requires_grad_(True)
        yy = O.lstm_cell_seq(g, lens_c, torch.eye(4 * H), w[d].cpu(), torch.zeros(4 * H), torch.zeros(4 * H), reverse=bool(d))
        (yy * dy.cpu()[:, :, d * H:(d + 1) * H]).sum().backward()
        assert float((y1[:, :, d * H:(d + 1) * H].cpu() - yy.detach()).abs().max()) < (6e-3 if fmt == 1 else 1.5e-3)
        assert float((d1[d].cpu() - g.grad).norm() / g.grad.norm()) < (5e-3 if fmt == 1 else 1.5e-3)


# ---------------------------------------------------------------- pack-by-length (compact) image GEMMs
def _valid_mask(T, B, lens):
    return (torch.arange(T)[:, None] < torch.tensor(lens)[None, :])          # [T,B]


@pytest.mark.parametrize("fmt", [1, 2])
def test_rowmap_and_compact_image(env, fmt):
    """ft_rowmap_build: batch-major compact rows, one separator per utterance (its first padded frame, or -1 when it has none),
    device-side row count; ft_bf16_image_rows: the image holds exactly those source rows (bit-identical to torch's cast), zero
    rows for -1 and up to ceil256(rows + 32), and its fused column sums are the sums over the mapped rows."""
    L, ops = env
    T, B, cols = 21, 5, 136
    lens = [21, 17, 9, 1, 0]
    lens32 = torch.tensor(lens, dtype=torch.int32, device="cuda")
    rm = ops.RowMap(lens32, T, B)
    want = []
    for b, l in enumerate(lens):
        want += [t * B + b for t in range(l)] + [l * B + b if l < T else -1]
    assert int(rm.rows.item()) == len(want) == sum(lens) + B and rm.cap == T * B + B
    assert rm.map[: len(want)].cpu().tolist() == want
    torch.manual_seed(3)
    src = torch.randn(T * B, cols, device="cuda")
    img = ops.Bf16Image(src, colsum=True, mode=fmt, rowmap=rm)
    Rz = (len(want) + 32 + 255) // 256 * 256
    raw = img.buf[: Rz * img.ld * 2].view(torch.int16).view(Rz, img.ld).cpu()
    ref = src.cpu().to(torch.bfloat16 if fmt == 1 else torch.float16).view(torch.int16)
    for i, r in enumerate(want):
        if r >= 0:
            assert torch.equal(raw[i, :cols], ref[r]), i
        else:
            assert int(raw[i].abs().max()) == 0
    assert int(raw[len(want):].abs().max()) == 0 and int(raw[:, cols:].abs().max()) == 0
    rows = [r for r in want if r >= 0]
    assert (img.colsum.cpu() - src.cpu()[rows].double().sum(0).float()).abs().max().item() < 1e-4


@pytest.mark.parametrize("act", [0, 1])
def test_compact_linear_equals_padded_linear_on_valid_rows(env, monkeypatch, act):
    """LinearFn with a RowMap (valid rows only, output rows scattered back) against the same call over all padded rows, bf16
    operands: forward outputs and input gradients of VALID rows are bit-identical (same per-row k order), weight / bias
    gradients agree to summation order -- provided the output gradient of padded rows is zero, as the masked losses make it;
    fill "y" reproduces the padded rows of the padded run (they all equal the utterance's first padded row), fill "dx" zeroes
    the input gradients of padded rows."""
    L, ops = env
    torch.manual_seed(21)
    T, B, K1, K2, N = 37, 5, 64, 48, 256
    lens = [37, 30, 12, 1, 5]
    lens32 = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m = _valid_mask(T, B, lens).cuda()
    x1 = torch.randn(T, B, K1, device="cuda") * m[..., None]              # padded frames of an utterance share one value (zeros here)
    x2 = torch.randn(T, B, K2, device="cuda") * m[..., None] + 0.25 * (~m)[..., None]
    W = torch.randn(N, K1 + K2, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    go = torch.randn(T, B, N, device="cuda") * m[..., None]
    res = []
    # the compact path with the two K pieces (FLOWTRON_GEMM_CAT=0) keeps the padded path's per-row k order: bit-identical; with the
    # concatenated image (the default since round 4: ONE K loop over K1 + K2) the same products meet in another order
    for rm, cat in ((None, True), (ops.RowMap(lens32, T, B), False), (ops.RowMap(lens32, T, B), True)):
        monkeypatch.setattr(ops, "_CAT_IMAGES", cat)
        d = [t.clone().requires_grad_(True) for t in (x1, x2, W, b)]
        out = ops.linear([d[0], d[1]], d[2], d[3], act=act, mode=1, rowmap=rm, fill="y+dx")
        out.backward(go)
        torch.cuda.synchronize()
        res.append([out.detach()] + [t.grad for t in d])
    (y0, dx10, dx20, dW0, db0), (y1, dx11, dx21, dW1, db1), (y2, dx12, dx22, dW2, db2) = res
    assert torch.equal(y0, y1)                                               # valid rows bit-identical, padded rows reproduced
    assert torch.equal(dx10[m], dx11[m]) and torch.equal(dx20[m], dx21[m])
    assert float(dx11[~m].abs().max()) == 0.0 and float(dx21[~m].abs().max()) == 0.0
    assert rel(dW1, dW0) < 2e-6 and rel(db1, db0) < 2e-6
    assert float((y2 - y0).abs().max()) <= 4e-6 * float(y0.abs().max())     # concatenated image: equal to fp32 summation order
    assert torch.equal(dx12[m], dx10[m]) and torch.equal(dx22[m], dx20[m])   # (dX never sums over the pieces)
    assert float(dx12[~m].abs().max()) == 0.0 and float(dx22[~m].abs().max()) == 0.0
    assert rel(dW2, dW0) < 2e-6 and rel(db2, db0) < 2e-6
    monkeypatch.setattr(ops, "_CAT_IMAGES", True)
    # unwritten rows stay untouched without a fill: poison them through the allocator and look
    d = [t.clone().requires_grad_(True) for t in (x1, x2, W, b)]
    out = ops.linear([d[0], d[1]], d[2], d[3], act=act, mode=1, rowmap=ops.RowMap(lens32, T, B), fill="")
    assert torch.equal(out[m], y2[m])                                        # (the concatenated-image path again)


@pytest.mark.parametrize("H", [128, 1024])
def test_compact_lstm_layer_equals_padded(env, H, monkeypatch):
    """ops.lstm_layer with a RowMap (compact input projection, compact dW_ih / dW_hh with the one-step shift as a one-ROW shift
    across the separator rows) against the padded path: y and dx bit-identical on valid frames, weight gradients to summation
    order.  H = 1024 runs the persistent recurrences, H = 128 the launch-per-step kernels."""
    L, ops = env
    monkeypatch.setattr(ops, "_GX16", False)          # (bitwise: gx as fp32 rows on both paths; the 16-bit rows are held against fp32 ones below)
    torch.manual_seed(5)
    T, B, K = 37, 5, 80
    lens = [37, 37, 20, 3, 1]
    lens32 = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m = _valid_mask(T, B, lens).cuda()
    x = torch.randn(T, B, K, device="cuda") * m[..., None]
    w_ih, w_hh = torch.randn(4 * H, K, device="cuda") * 0.1, torch.randn(4 * H, H, device="cuda") / H ** 0.5
    b_ih, b_hh = torch.randn(4 * H, device="cuda") * 0.1, torch.randn(4 * H, device="cuda") * 0.1
    go = torch.randn(T, B, H, device="cuda") * m[..., None]
    res = []
    for rm in (None, ops.RowMap(lens32, T, B)):
        d = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        y = ops.lstm_layer(d[0], lens32, d[1], d[2], d[3], d[4], mode=1, rowmap=rm, fill="dx")
        y.backward(go)
        torch.cuda.synchronize()
        res.append([y.detach()] + [t.grad for t in d])
    ops.check_persist_status()
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1][m], res[1][1][m]) and float(res[1][1][~m].abs().max()) == 0.0
    for a, b_, name in zip(res[1][2:], res[0][2:], ("dW_ih", "dW_hh", "db_ih", "db_hh")):
        assert rel(a, b_) < 5e-6, (name, rel(a, b_))
    if H == 1024:
        # round 6: the compact projection hands gx to the persistent recurrence as 16-BIT rows (ops.gx16_ok): one more operand-grade
        # rounding (2^-9 relative on gx) -- y within 2e-2 absolute, gradients within 2e-2 relative of the fp32-row path
        monkeypatch.setattr(ops, "_GX16", True)
        d = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        y = ops.lstm_layer(d[0], lens32, d[1], d[2], d[3], d[4], mode=1, rowmap=ops.RowMap(lens32, T, B), fill="dx")
        y.backward(go)
        torch.cuda.synchronize()
        ops.check_persist_status()
        assert mad(y, res[1][0]) < 2e-2 and mad(y, res[1][0]) > 0.0, "the 16-bit rows were not used"
        for a, b_, name in zip([t.grad for t in d], res[1][1:], ("dx", "dW_ih", "dW_hh", "db_ih", "db_hh")):
            assert rel(a, b_) < 2e-2, (name, rel(a, b_))


@pytest.mark.parametrize("H,gate", [(512, False), (640, True)])
def test_hidden_size_below_1024_runs_on_the_persistent_kernels(env, H, gate, monkeypatch):
    """A layer with H < 1024 hidden units (the reference's nn.LSTM takes any n_hidden, flowtron.py:654-655) runs as its zero-padded
    1024-unit twin on the persistent kernels (ops.lstm_pad_width / pad_gate_blocks) instead of the launch-per-step kernels: outputs,
    the gate-layer values riding on the projection, and every gradient against the unpadded launch-per-step path (_PAD_H = False) in
    the same operand format -- the two differ by the summation order over k and by the 16-bit gx rows of the persistent path."""
    L, ops = env
    torch.manual_seed(6)
    T, B, K = 96, 6, 80
    lens = [96, 96, 70, 33, 2, 1]
    lens32 = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m = _valid_mask(T, B, lens).cuda()
    x = torch.randn(T, B, K, device="cuda") * m[..., None]
    w_ih, w_hh = torch.randn(4 * H, K, device="cuda") * 0.1, torch.randn(4 * H, H, device="cuda") / H ** 0.5
    b_ih, b_hh = torch.randn(4 * H, device="cuda") * 0.1, torch.randn(4 * H, device="cuda") * 0.1
    gw, gb = torch.randn(1, K, device="cuda") * 0.1, torch.randn(1, device="cuda")
    go = torch.randn(T, B, H, device="cuda") * m[..., None]
    assert ops.lstm_pad_width(B, H, False, 1, x.device, T)
    res = []
    for pad in (False, True):
        monkeypatch.setattr(ops, "_PAD_H", pad)
        d = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh, gw, gb)]
        out = ops.lstm_layer(d[0], lens32, d[1], d[2], d[3], d[4], mode=1, rowmap=ops.RowMap(lens32, T, B), fill="dx",
                             gate=(d[5], d[6]) if gate else None)
        y, g = out if gate else (out, None)
        assert y.shape == (T, B, H) and y.is_contiguous()
        loss = (y * go).sum() + (g[m].sum() if gate else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        res.append([y.detach()] + ([g.detach()[m]] if gate else []) + [t.grad for t in d[:5 + 2 * int(gate)]])
    ops.check_persist_status()
    assert float((res[1][0] * (~m)[..., None]).abs().max()) == 0.0                  # padded frames stay zero
    assert mad(res[1][0], res[0][0]) < 2e-2 and mad(res[1][0], res[0][0]) > 0.0, "the padded persistent path was not taken"
    names = (["gate"] if gate else []) + ["dx", "dW_ih", "dW_hh", "db_ih", "db_hh"] + (["dgate_w", "dgate_b"] if gate else [])
    for a, b_, name in zip(res[1][1:], res[0][1:], names):
        assert rel(a, b_) < 2e-2, (name, rel(a, b_))


@pytest.mark.parametrize("act", [0, 1])
def test_wide_stage_image_gemm_two_inputs_is_bit_identical_to_the_32_wide_stages(env, act, monkeypatch):
    """The single-buffer 64-wide-stage kernel (gemm_bf16_k<.., 64, true>, the default wherever K is a whole number of 64-wide stages)
    against the 32-wide stages (FT_GEMM_BF16_WIDE=0) on a Linear over TWO inputs (one GEMM over the concatenated image, K = 320 + 256),
    4800 rows with ragged lengths, padded and compact row space, bias + activation: the accumulation order per output element is
    the same (32-wide k steps in order), so the outputs must be BIT-identical -- any staging / swizzle / barrier slip shows."""
    L, ops = env
    torch.manual_seed(77)
    T, B, K1, K2, N = 150, 32, 320, 256, 640
    lens = [150 - 3 * i for i in range(B)]
    lens32 = torch.tensor(lens, dtype=torch.int32, device="cuda")
    x1, x2 = torch.randn(T, B, K1, device="cuda"), torch.randn(T, B, K2, device="cuda")
    W = torch.randn(N, K1 + K2, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    res = {}
    for wide in ("0", "1"):
        monkeypatch.setenv("FT_GEMM_BF16_WIDE", wide)
        outs = []
        for rm in (None, ops.RowMap(lens32, T, B)):
            outs.append(ops.linear([x1, x2], W, b, act=act, mode=1, rowmap=rm, fill="y").detach().clone())
        res[wide] = outs
    for a_, b_ in zip(res["0"], res["1"]):
        assert torch.equal(a_, b_), (a_ - b_).abs().max().item()
    ref = torch.cat([x1, x2], 2).bfloat16().float() @ W.bfloat16().float().t() + b
    ref = torch.tanh(ref) if act else ref
    assert mad(res["1"][0], ref) < 2e-3


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("K", [256, 448, 1664])
def test_wide_stage_image_gemm_is_bit_identical_to_the_32_wide_stages(env, K, fmt, monkeypatch):
    """gemm_bf16_k<.., 64, true> (64-wide k stages in one LDS buffer: whole-line DMA pieces for the k-contiguous operands, a k-major
    operand as two of its 32-wide sub-stages) against the 32-wide stages (FT_GEMM_BF16_WIDE=0): same k order per output element, so
    forward (both operands k-contiguous, padded and compact rows, bias + tanh) and input gradient (k-major weight image) must be
    BIT-identical.  K = 448: 7 whole 64-wide stages;  K = 1664: the decoder projection's own width.  fmt 1 / 2: bf16 / fp16 operands
    (the two builds of the same source)."""
    L, ops = env
    torch.manual_seed(78)
    rnd = (lambda t: t.bfloat16().float()) if fmt == 1 else (lambda t: t.half().float())
    T, B, N = 37, 8, 320
    lens32 = torch.tensor([37 - 4 * i for i in range(B)], dtype=torch.int32, device="cuda")
    x = torch.randn(T, B, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    dy = torch.randn(T * B, N, device="cuda")
    res = {}
    for wide in ("0", "1"):
        monkeypatch.setenv("FT_GEMM_BF16_WIDE", wide)
        outs = []
        for rm in (None, ops.RowMap(lens32, T, B)):
            outs.append(ops.linear([x], W, b, act=1, mode=fmt, rowmap=rm, fill="y").detach().clone())
        # dX[M, K] = dy[M, N] . W[N, K]: A k-contiguous (reduction over N), B = the weight image read k-major
        d_img, w_img = ops.Bf16Image(dy, mode=fmt), ops.Bf16Image(W, mode=fmt)
        dx = torch.empty(T * B, K, device="cuda")
        ops.gemm_img(d_img, 0, d_img.ptr(), w_img, 1, w_img.ptr(), dx, T * B, K, N, K)
        outs.append(dx)
        res[wide] = outs
    for a_, b_ in zip(res["0"], res["1"]):
        assert torch.equal(a_, b_), (a_ - b_).abs().max().item()
    ref = torch.tanh(rnd(x) @ rnd(W).t() + b)
    assert mad(res["1"][0], ref) < 2e-3
    assert mad(res["1"][2], rnd(dy) @ rnd(W)) < 2e-2


# ---------------------------------------------------------------- cumulative attention, fused frames (csrc/cumm_fused.hip)
def _cumm_reference(Q, V, text, wk, v, w1, b1, w2, b2, in_lens, temp):
    """flowtron.py:697-723 + :129-152 + :544-592 in float64 torch (the per-frame loop of the reference), with autograd."""
    T, B, A = Q.shape
    Lk = text.shape[0]
    cumm = Q.new_zeros(B, 1, Lk)
    prev = Q.new_zeros(B, 1, Lk)
    mask = torch.arange(Lk)[None, :] >= in_lens[:, None]
    ctxs, attns, lps = [], [], []
    for i in range(T):
        x = torch.cat([cumm, prev], 1)
        h = torch.relu(F.conv1d(x, w1, b1, padding=2))
        c = torch.sigmoid(F.conv1d(h, w2, b2, padding=1)).permute(2, 0, 1)          # [L,B,E]
        K = (text * c) @ wk.t()                                                        # [L,B,A]
        e = (torch.tanh(Q[i][None] + K) @ v / temp).t().masked_fill(mask, -float("inf"))
        p = torch.softmax(e, 1)
        ctxs.append(torch.einsum("bl,lba->ba", p, V))
        attns.append(p)
        lps.append(torch.log(p + 1e-8))
        prev = p[:, None, :]
        cumm = cumm + prev
    return torch.stack(ctxs, 0), torch.stack(attns, 1), torch.stack(lps, 1)


@pytest.mark.parametrize("fmt,split", [(1, "1"), (2, "1"), (1, "0"), (1, "chunk4"), (1, "frames")])
@pytest.mark.parametrize("T,Lk,lens", [(23, 57, [57, 52, 26, 7]), (5, 157, [157, 130, 33]), (9, 20, [20, 1])])
def test_fused_cumulative_attention_frames_vs_float64_reference(env, monkeypatch, fmt, split, T, Lk, lens):
    """ONE fused launch per frame and direction (ft_cumm_attn_fwd / _bwd -> csrc/cumm_fused.hip, the path of
    the 16-bit operand modes at the config.json width E = A = 640) against the reference's per-frame loop in float64 torch, and
    beside it the launch chain it replaces (FT_CUMM_FUSED=0) in the same operand format: forward outputs and every gradient.  The
    lengths hit the tile edges of both kernels: in_len == L, a multiple of the backward's 26 own rows (52, 26), of the forward's
    32 (rows beyond it exit), a single text position.  Bar: the fused path is as close to float64 as the chain (<= 2x its
    deviation + a floor at operand-rounding level), and both within the 16-bit tolerance written below."""
    L, ops = env
    E = A = 640
    B = len(lens)
    gen = torch.Generator().manual_seed(100 + T + Lk)
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    Q, V, text = rn(T, B, A) * 0.7, rn(Lk, B, A), rn(Lk, B, E) * 0.7
    wk, v = rn(A, E) / E ** 0.5 * 1.5, rn(A) / A ** 0.5 * 6.0
    w1, b1, w2, b2 = rn(32, 2, 5) * 0.6, rn(32) * 0.3, rn(E, 32, 3) * 0.25, rn(E) * 0.3
    in_lens = torch.tensor(lens)
    temp = 0.9
    gc_, ga_, gl_ = rn(T, B, A), rn(B, T, Lk), rn(B, T, Lk) * 0.1
    valid = (torch.arange(Lk)[None, :] < in_lens[:, None])[:, None, :].double()
    leaves = [t.clone().requires_grad_(True) for t in (Q, V, text, wk, v, w1, b1, w2, b2)]
    rc, ra, rl = _cumm_reference(*leaves, in_lens, temp)
    ((rc * gc_).sum() + (ra * ga_).sum() + (rl * gl_ * valid).sum()).backward()
    ref_out = [rc.detach(), ra.detach(), (rl * valid).detach()]
    ref_grad = [t.grad for t in leaves]

    def run(fused):
        monkeypatch.setenv("FT_CUMM_FUSED", "1" if fused else "0")
        # split: forward tiles as two column-half workgroups (default when they fit the chip) / one; chunk4: the backward's streams in
        # chunks of four frames -- several accumulated rounds of the weight-gradient GEMMs, a partial top chunk
        # default: the frames of a pass inside PERSISTENT launches (granule hand-offs); frames: one launch per frame
        monkeypatch.setenv("FLOWTRON_CUMM_PERSIST", "0" if split == "frames" else "1")
        monkeypatch.setenv("FT_CUMM_SPLIT", "0" if split == "0" else "1")
        if split.startswith("chunk"):
            monkeypatch.setenv("FT_CUMM_CHUNK", "4")
        lv = [t.float().cuda().requires_grad_(True) for t in (Q, V, text, wk, v.reshape(1, -1), w1, b1, w2, b2)]
        c, a_, lp = ops.CummAttnSeqFn.apply(*lv, in_lens.int().cuda(), temp, fmt)
        ((c * gc_.float().cuda()).sum() + (a_ * ga_.float().cuda()).sum() + (lp * (gl_ * valid).float().cuda()).sum()).backward()
        torch.cuda.synchronize()
        outs = [c.detach().cpu().double(), a_.detach().cpu().double(), (lp.detach().cpu().double() * valid)]
        return outs, [t.grad.detach().cpu().double().reshape(r.shape) for t, r in zip(lv, ref_grad)]

    names = ["Q", "V", "text", "w_key", "v", "w1", "b1", "w2", "b2"]
    fo, fg = run(True)
    co, cg = run(False)
    report = []
    for nm, f, c_, r in zip(["ctx", "attn", "logprob"], fo, co, ref_out):
        ef, ec = (f - r).abs().max().item(), (c_ - r).abs().max().item()
        report.append((nm, ef, ec))
        assert ef <= 2.0 * ec + 4e-3 and ef < 6e-2, report
    for nm, f, c_, r in zip(names, fg, cg, ref_grad):
        nr = r.norm().item() + 1e-30
        ef, ec = (f - r).norm().item() / nr, (c_ - r).norm().item() / nr
        report.append(("d" + nm, ef, ec))
        assert ef <= 2.0 * ec + 5e-3 and ef < 6e-2, report
    print("\n[fused cumulative attention fmt %d T %d L %d] (name, fused vs f64, chain vs f64): %s" % (fmt, T, Lk, report))
