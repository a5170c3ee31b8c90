"""Round 6 (-m gpu): the persistent recurrences with rows per XCD group R = 4 / 8 / 16, time windows with carried state and two
roles per launch (csrc/lstm_roles.hip, through the C ABI: ft_lstm_roles_*) against the launch-per-step kernels (ft_lstm_seq_fwd /
_bwd, which tests/test_gpu_ops.py::test_lstm_seq* hold against the CPU oracle), and the decoder layer pair pipeline
(ops.DecoderPairFn) against two single-layer calls.  What nn.LSTM over packed sequences computes: flowtron.py:654-655, 689-694.
Forward: bit-identical for every geometry; backward: bit-identical across windowings / role placements of one R, fp32 rounding
(different association of the 32 producers' partial sums) across R and against the launch-per-step kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 1024


@pytest.fixture(scope="module")
def env():
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    assert torch.cuda.is_available(), "these tests need the MI355X"
    if not ops.persist_usable(torch.device("cuda", 0)):
        pytest.skip("persistent kernels not usable on this device")
    return L, ops


def make(T, B, seed, ragged=True):
    torch.manual_seed(seed)
    gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
    w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
    lens = torch.randint(max(1, T // 3), T + 1, (B,), dtype=torch.int32) if ragged else torch.full((B,), T, dtype=torch.int32)
    lens[seed % B] = T
    if ragged and B > 3:
        lens[(seed + 1) % B] = 1                         # an utterance that ends inside the first window
    return gx, w, lens.cuda()


def bufs(T, B):
    return torch.full((T, B, H), 7.0, device="cuda"), torch.full((T, B, 4 * H), 7.0, device="cuda"), torch.full((T, B, H), 7.0, device="cuda")


def step_fwd(L, gx, w, lens, mode):
    T, B = gx.shape[:2]
    y, g, c = bufs(T, B)
    for b0 in range(0, B, 64):                                 # (the launch-per-step kernels take 64 rows)
        nb = min(64, B - b0)
        work = torch.empty(L.lib().ft_lstm_workspace_bytes(nb, H), device="cuda", dtype=torch.uint8)
        ys, gs, cs = bufs(T, nb)
        gxs, ls = gx[:, b0:b0 + nb].contiguous(), lens[b0:b0 + nb].contiguous()      # (named: a temporary would be recycled before the launch)
        L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gxs), L.ptr(w), L.ptr(ls), L.ptr(ys), H, L.ptr(gs), L.ptr(cs), L.ptr(work), T, nb, H, 0, mode, L.stream()),
                "ft_lstm_seq_fwd")
        y[:, b0:b0 + nb], g[:, b0:b0 + nb], c[:, b0:b0 + nb] = ys, gs, cs
    return y, g, c


def step_bwd(L, dy, w, lens, g, c, mode):
    T, B = dy.shape[:2]
    dgx = torch.empty(T, B, 4 * H, device="cuda")
    for b0 in range(0, B, 64):
        nb = min(64, B - b0)
        work = torch.empty(L.lib().ft_lstm_workspace_bytes(nb, H), device="cuda", dtype=torch.uint8)
        d = torch.empty(T, nb, 4 * H, device="cuda")
        dys, ls, gs, cs = dy[:, b0:b0 + nb].contiguous(), lens[b0:b0 + nb].contiguous(), g[:, b0:b0 + nb].contiguous(), c[:, b0:b0 + nb].contiguous()
        L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dys), H, L.ptr(w), L.ptr(ls), L.ptr(gs), L.ptr(cs), L.ptr(d), L.ptr(work), T, nb, H, 0, mode, L.stream()),
                "ft_lstm_seq_bwd")
        dgx[:, b0:b0 + nb] = d
    return dgx


def same_fwd(lens, a, b):
    T = a[0].shape[0]
    m = torch.arange(T, device="cuda")[:, None] < lens[None, :]
    return bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1][m], b[1][m])) and bool(torch.equal(a[2][m], b[2][m]))


def clean(ops):
    assert ops.check_persist_status(), "a persistent launch timed out"


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("B,R", [(32, 4), (7, 4), (32, 8), (20, 8), (64, 8), (128, 16), (100, 16), (40, 16)])
def test_forward_rows_per_group_bit_identical_to_launch_per_step(env, B, R, fmt):
    """one role, any R: y / saved gates / saved cell of every valid (t, b) bit-identical to ft_lstm_seq_fwd, zeros on pad rows;
    partially filled groups (B not a multiple of R) and groups without rows included"""
    L, ops = env
    if fmt == 2 and (B, R) not in ((32, 4), (64, 8), (100, 16)):
        pytest.skip("fp16 twins: one case per R")
    T = 37
    gx, w, lens = make(T, B, 11 * B + R)
    ref = step_fwd(L, gx, w, lens, fmt)
    out = bufs(T, B)
    wimg = ops.roles_wimg(w, fmt, False)
    ops.roles_launch([ops.fwd_role(gx, lens, out[0], out[1], out[2], wimg)], R, fmt, gx.device)
    torch.cuda.synchronize()
    clean(ops)
    assert same_fwd(lens, ref, out)
    pad = ~(torch.arange(T, device="cuda")[:, None] < lens[None, :])
    assert float(out[0][pad].abs().max()) == 0.0


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("B,R,edges", [(32, 4, [0, 70]), (32, 8, [0, 33, 70]), (20, 8, [0, 70]), (100, 16, [0, 9, 70])])
def test_forward_takes_16bit_gx_rows(env, B, R, edges, fmt):
    """gx as 16-BIT rows of the operand format (what the projection GEMM writes with FT_GEMM_C16; ft_lstm_fwd_role.gx16): widened
    exactly, so the launch equals the launch-per-step kernel fed with the same values in fp32 -- bit for bit, over several bursts of
    the doubled burst length, with windows and partially filled groups"""
    L, ops = env
    T = edges[-1]
    gx, w, lens = make(T, B, 7 * B + R + fmt)
    gx16 = gx.to(ops.op16_dtype(fmt))
    ref = step_fwd(L, gx16.float(), w, lens, fmt)
    out = bufs(T, B)
    st = torch.zeros(2, B, H, device="cuda")
    wimg = ops.roles_wimg(w, fmt, False)
    for k in range(len(edges) - 1):
        ops.roles_launch([ops.fwd_role(gx16, lens, out[0], out[1], out[2], wimg, edges[k], edges[k + 1], st)], R, fmt, gx.device)
    torch.cuda.synchronize()
    clean(ops)
    assert same_fwd(lens, ref, out)


def test_image_gemm_writes_16bit_rows(env):
    """ft_gemm_img with FT_GEMM_C16: the fp32 result + bias rounded ONCE to the operand format in the epilogue, through a row map
    (compact rows scattered to their time-major places) -- equal to rounding the fp32 output of the same GEMM"""
    L, ops = env
    T, B, K, N = 40, 6, 96, 256
    torch.manual_seed(2)
    lens = torch.tensor([40, 33, 1, 17, 40, 8], dtype=torch.int32, device="cuda")
    x = torch.randn(T, B, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.2
    bias = torch.randn(N, device="cuda")
    for fmt in (1, 2):
        rm = ops.RowMap(lens, T, B)
        xi, wi = ops.Bf16Image(x.reshape(T * B, K), mode=fmt, rowmap=rm), ops.Bf16Image(W, mode=fmt)
        y32 = torch.zeros(T, B, N, device="cuda")
        y16 = torch.zeros(T, B, N, device="cuda", dtype=ops.op16_dtype(fmt))
        ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y32, rm.cap, N, K, N, bias=bias, rowmap=rm, compact=1)
        ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y16, rm.cap, N, K, N, bias=bias, rowmap=rm, compact=1, c16=True)
        torch.cuda.synchronize()
        valid = torch.arange(T, device="cuda")[:, None] < lens[None, :]
        assert torch.equal(y16[valid], y32[valid].to(y16.dtype))


@pytest.mark.parametrize("R", [4, 8])
@pytest.mark.parametrize("edges", [[0, 13, 14, 40, 61], [0, 61], [0, 1, 60, 61]])
def test_forward_windows_and_two_roles_bit_identical(env, R, edges):
    """a sequence walked in windows with carried (h, c) equals one launch; two different recurrences as two roles of one launch
    equal one launch each -- including the skewed pipeline DecoderPairFn issues (role 0 on window k, role 1 on window k - 1)"""
    L, ops = env
    T, B, mode = 61, 32, L.FT_BF16
    a, b = make(T, B, 5), make(T, B, 6)
    ra, rb = step_fwd(L, *a, mode), step_fwd(L, *b, mode)
    wa, wb = ops.roles_wimg(a[1], mode, False), ops.roles_wimg(b[1], mode, False)
    oa = bufs(T, B)
    st = torch.zeros(2, B, H, device="cuda")
    for k in range(len(edges) - 1):
        ops.roles_launch([ops.fwd_role(a[0], a[2], oa[0], oa[1], oa[2], wa, edges[k], edges[k + 1], st)], R, mode, a[0].device)
    torch.cuda.synchronize()
    clean(ops)
    assert same_fwd(a[2], ra, oa)
    if R == 8:
        oa, ob = bufs(T, B), bufs(T, B)
        sa, sb = torch.zeros(2, B, H, device="cuda"), torch.zeros(2, B, H, device="cuda")
        n = len(edges) - 1
        for k in range(n + 1):
            roles = []
            if k < n:
                roles.append(ops.fwd_role(a[0], a[2], oa[0], oa[1], oa[2], wa, edges[k], edges[k + 1], sa))
            if k > 0:
                roles.append(ops.fwd_role(b[0], b[2], ob[0], ob[1], ob[2], wb, edges[k - 1], edges[k], sb))
            ops.roles_launch(roles, 8 if len(roles) == 2 else 4, mode, a[0].device)
        torch.cuda.synchronize()
        clean(ops)
        assert same_fwd(a[2], ra, oa) and same_fwd(b[2], rb, ob)


@pytest.mark.parametrize("fmt,tol", [(1, 1e-3), (2, 2e-4)])
@pytest.mark.parametrize("B,R", [(32, 4), (32, 8), (20, 8), (64, 8), (100, 16)])
def test_backward_rows_per_group_matches_launch_per_step(env, B, R, fmt, tol):
    """reduce-scatter backward at any R against ft_lstm_seq_bwd on the saved tensors of a real forward: same 16-bit operand rounding of
    dgates, another fp32 association of the recurrent product (rel-L2 as for lstm_persist_bwd_rs_k: observed 2.5e-4 at T 862); the
    first step (no recurrent term) to a few ulps; windows with carried (dgates, dc) and two roles bit-identical to one launch of the
    same R; the compact image output = ft_bf16_image_rows of the fp32 rows"""
    L, ops = env
    if fmt == 2 and (B, R) != (32, 8):
        pytest.skip("fp16 twin: one case")
    T = 45
    gx, w, lens = make(T, B, 3 * B + R)
    y, g, c = step_fwd(L, gx, w, lens, fmt)
    torch.manual_seed(1)
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    ref = step_bwd(L, dy, w, lens, g, c, fmt)
    wimg = ops.roles_wimg(w, fmt, True)
    d1 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    ops.roles_launch([ops.bwd_role(dy, lens, g, c, d1, wimg)], R, fmt, dy.device, backward=True)
    torch.cuda.synchronize()
    clean(ops)
    assert float((d1 - ref).norm() / ref.norm()) <= tol
    last = int(lens.max()) - 1
    rows = lens == last + 1
    assert float((d1[last][rows] - ref[last][rows]).abs().max()) <= 2e-6 * float(ref[last][rows].abs().max()) + 1e-9
    pad = ~(torch.arange(T, device="cuda")[:, None] < lens[None, :])
    assert float(d1[pad].abs().max()) == 0.0
    # windows + carried state
    d2 = torch.full((T, B, 4 * H), 7.0, device="cuda")
    st = (torch.zeros(B, 4 * H, device="cuda"), torch.zeros(B, H, device="cuda"))
    edges = [0, 9, 10, 31, T]
    for k in reversed(range(len(edges) - 1)):
        ops.roles_launch([ops.bwd_role(dy, lens, g, c, d2, wimg, edges[k], edges[k + 1], st, carry_in=k < len(edges) - 2)], R, fmt, dy.device, backward=True)
    torch.cuda.synchronize()
    clean(ops)
    assert torch.equal(d1, d2)
    if B <= 32:
        # image-only output and, at R = 8, a second role beside it
        rm = ops.row_map(lens, T, B)
        img = ops.Bf16Image.empty_rows(4 * H, rm, fmt, dy.device)
        roles = [ops.bwd_role(dy, lens, g, c, None, wimg, dimg=img)]
        d3 = torch.empty(T, B, 4 * H, device="cuda")
        if R == 8:
            roles.append(ops.bwd_role(dy, lens, g, c, d3, wimg))
        ops.roles_launch(roles, R, fmt, dy.device, backward=True)
        ref_img = ops.Bf16Image(d1.reshape(T * B, 4 * H), colsum=True, mode=fmt, rowmap=rm)
        torch.cuda.synchronize()
        clean(ops)
        nbytes = int((lens.sum() + B).item()) * img.ld * 2
        assert torch.equal(img.buf[:nbytes], ref_img.buf[:nbytes])
        assert float((img.colsum - ref_img.colsum).norm() / ref_img.colsum.norm()) <= 1e-5
        if R == 8:
            assert torch.equal(d3, d1)


def test_backward_four_rows_bit_identical_to_round5_kernel(env):
    """R = 4 keeps lstm_persist_bwd_rs_k's summation order: bit-identical to ft_lstm_persist_bwd (transport 21)"""
    L, ops = env
    T, B, mode = 33, 32, L.FT_BF16
    gx, w, lens = make(T, B, 77)
    y, g, c = step_fwd(L, gx, w, lens, mode)
    dy = torch.randn(T, B, H, device="cuda") * 0.1
    d0 = torch.empty(T, B, 4 * H, device="cuda")
    work = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
    L.check(L.lib().ft_lstm_persist_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g), L.ptr(c), L.ptr(d0), L.ptr(work), L.ptr(ops.persist_status(dy.device)),
                                        T, B, H, 21, L.stream()), "ft_lstm_persist_bwd")
    d1 = torch.empty(T, B, 4 * H, device="cuda")
    ops.roles_launch([ops.bwd_role(dy, lens, g, c, d1, ops.roles_wimg(w, mode, True))], 4, mode, dy.device, backward=True)
    torch.cuda.synchronize()
    clean(ops)
    assert torch.equal(d0, d1)


@pytest.mark.parametrize("nch,nch_bwd", [(3, 0), (4, -1), (2, 3)])
def test_decoder_pair_pipeline_equals_two_single_layer_calls(env, monkeypatch, nch, nch_bwd):
    """ops.decoder_pair (layer 0's projection + DecoderPairFn: the two decoder layers as a chunk pipeline of role launches, layer 1's
    input projection per chunk in between) against two ops.lstm_layer calls on the same parameters (nn.LSTM(1664, 1024, 2),
    flowtron.py:654-655, 760-765): h bit-identical (the chunk GEMMs multiply the same rows with the same images), gate logits
    identical, every gradient to the backward recurrence's fp32 rounding -- with the sequential backward, a backward pipeline of the
    forward's chunks and one of its own"""
    L, ops = env
    import torch.nn as nn
    T, B, A, mode = 48, 32, 640, L.FT_BF16
    torch.manual_seed(3)
    p = nn.LSTM(H + A, H, 2).cuda()
    gw, gb = (torch.randn(1, H + A, device="cuda") * 0.02).requires_grad_(True), torch.zeros(1, device="cuda", requires_grad=True)
    lens = torch.randint(10, T + 1, (B,), dtype=torch.int32)
    lens[3] = T
    lens = lens.cuda()
    x0 = torch.randn(T, B, H, device="cuda") * 0.3
    c0 = torch.randn(T, B, A, device="cuda") * 0.3
    dh = torch.randn(T, B, H, device="cuda") * 0.1
    dg = torch.randn(T, B, 1, device="cuda") * 0.1
    valid = (torch.arange(T, device="cuda")[:, None] < lens[None, :])[..., None].float()

    def run(pair):
        x, cx = x0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        for q in list(p.parameters()) + [gw, gb]:
            q.grad = None
        rm = ops.row_map(lens, T, B)
        if pair:
            h, gates = ops.decoder_pair(x, lens, p, mode, [cx], rm, "dx", (gw, gb), nch)
        else:
            h0, gates = ops.lstm_layer(x, lens, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0, mode=mode, xs_extra=[cx], rowmap=rm,
                                       fill="dx", gate=(gw, gb))
            h = ops.lstm_layer(h0, lens, p.weight_ih_l1, p.weight_hh_l1, p.bias_ih_l1, p.bias_hh_l1, mode=mode, rowmap=rm)
        ((h * dh).sum() + (gates * dg * valid).sum()).backward()
        torch.cuda.synchronize()
        clean(ops)
        grads = {n: q.grad.clone() for n, q in p.named_parameters()}
        grads.update(gw=gw.grad.clone(), gb=gb.grad.clone(), x=(x.grad * valid).clone(), cx=(cx.grad * valid).clone())
        return h.detach(), gates.detach(), grads

    monkeypatch.setattr(ops, "_PAIR_CHUNKS_BWD", nch_bwd)
    h_ref, g_ref, gr = run(False)
    h_p, g_p, gp = run(True)
    assert torch.equal(h_ref, h_p)
    assert torch.equal(g_ref * valid, g_p * valid)
    for n in gr:
        e = float((gp[n] - gr[n]).norm() / (gr[n].norm() + 1e-30))
        assert e <= 2e-3, (n, e)
