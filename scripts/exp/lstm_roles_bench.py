"""Round 6 probe: the persistent recurrences with R = 4 / 8 / 16 rows per XCD group, windows and two roles (csrc/lstm_roles.hip)
against the launch-per-step kernels (csrc/lstm.hip).  Prints us/step and equality for:
  forward / backward, one role: R 4 (B 32), R 8 (B 32 on four XCDs; B 64), R 16 (B 128)
  windows: the B 32 sequence in chunks with carried state vs one launch
  two roles: two different recurrences in one launch vs one launch each
Run on the GPU box: python scripts/exp/lstm_roles_bench.py   (T=862 by default; CHECK_ONLY=1 skips the timing loops)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
from flowtron_amd import ops

T, H = int(os.environ.get("T", 862)), 1024
dev = torch.device("cuda", 0)
mode = L.FT_BF16
REPS = int(os.environ.get("REPS", 5))
res = {}


def bench_lens(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    l = torch.randint(int(T * 0.45), T + 1, (B,), generator=g, dtype=torch.int32)
    l[torch.randint(0, B, (1,), generator=g)] = T
    return l.to(dev)


def timeit(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / T)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def make(B, seed):
    torch.manual_seed(seed)
    gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
    w = torch.randn(4 * H, H, device=dev) * (1.0 / 32)
    return gx, w, bench_lens(B, seed)


def bufs(B):
    return (torch.full((T, B, H), 7.0, device=dev), torch.full((T, B, 4 * H), 7.0, device=dev), torch.full((T, B, H), 7.0, device=dev))


status = ops.persist_status(dev)


def old_fwd(gx, w, lens, y, g, c):
    """the yardstick: the launch-per-step kernels (csrc/lstm.hip), batch chunks of 64 rows.  (The committed
    profiles/r06_persist_rows_per_group.log was measured against the round-5 persistent kernel lstm_persist_fwd_k, 1.76 us per
    step, which round 6 then removed.)"""
    B = gx.shape[1]
    for b0 in range(0, B, 64):
        nb = min(64, B - b0)
        work = torch.empty(L.lib().ft_lstm_workspace_bytes(nb, H), device=dev, dtype=torch.uint8)
        gxs, ls = gx[:, b0:b0 + nb].contiguous(), lens[b0:b0 + nb].contiguous()
        ys, gs, cs = torch.empty(T, nb, H, device=dev), torch.empty(T, nb, 4 * H, device=dev), torch.empty(T, nb, H, device=dev)
        L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gxs), L.ptr(w), L.ptr(ls), L.ptr(ys), H, L.ptr(gs), L.ptr(cs), L.ptr(work), T, nb, H, 0, mode, L.stream()), "step fwd")
        y[:, b0:b0 + nb], g[:, b0:b0 + nb], c[:, b0:b0 + nb] = ys, gs, cs


def old_bwd(dy, w, lens, g, c, dgx):
    B = dy.shape[1]
    for b0 in range(0, B, 64):
        nb = min(64, B - b0)
        work = torch.empty(L.lib().ft_lstm_workspace_bytes(nb, H), device=dev, dtype=torch.uint8)
        dys, ls, gs, cs = dy[:, b0:b0 + nb].contiguous(), lens[b0:b0 + nb].contiguous(), g[:, b0:b0 + nb].contiguous(), c[:, b0:b0 + nb].contiguous()
        d = torch.empty(T, nb, 4 * H, device=dev)
        L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dys), H, L.ptr(w), L.ptr(ls), L.ptr(gs), L.ptr(cs), L.ptr(d), L.ptr(work), T, nb, H, 0, mode, L.stream()), "step bwd")
        dgx[:, b0:b0 + nb] = d


def act_mask(lens):
    return torch.arange(T, device=dev)[:, None] < lens[None, :]


def same_fwd(lens, a, b):
    m = act_mask(lens)
    return bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1][m], b[1][m])) and bool(torch.equal(a[2][m], b[2][m]))


def st():
    v = int(status.item()); status.zero_(); return v


print("T = %d, bf16 operands; us per step = launch time / T" % T, flush=True)
# ---------------------------------------------------------------- forward, one role
for B, R in ((32, 4), (32, 8), (64, 8), (128, 16), (64, 16)):
    gx, w, lens = make(B, B + R)
    ref = bufs(B); old_fwd(gx, w, lens, *ref)
    wimg = ops.roles_wimg(w, mode, False)
    out = bufs(B)
    run = lambda: ops.roles_launch([ops.fwd_role(gx, lens, out[0], out[1], out[2], wimg)], R, mode, dev)
    run(); torch.cuda.synchronize()
    eq, s_ = same_fwd(lens, ref, out), st()
    line = "fwd  B %3d R %2d: bit-identical to the launch-per-step kernels %s  status %d" % (B, R, eq, s_)
    if not os.environ.get("CHECK_ONLY"):
        t_new = timeit(run); t_old = timeit(lambda: old_fwd(gx, w, lens, *ref))
        res["fwd_B%d_R%d" % (B, R)] = {"us_per_step": t_new[0], "median": t_new[1], "launch_per_step_us_per_step": t_old[0], "identical": eq}
        line += "  | %.3f us/step (median %.3f) against %.3f for the launch-per-step kernels (%d x 64 rows)" % (t_new[0], t_new[1], t_old[0], (B + 63) // 64)
    print(line, flush=True)

# ---------------------------------------------------------------- forward, windows (B 32, R 4 and R 8) and two roles
gx, w, lens = make(32, 1)
gx2, w2, lens2 = make(32, 2)
ref, ref2 = bufs(32), bufs(32)
old_fwd(gx, w, lens, *ref); old_fwd(gx2, w2, lens2, *ref2)
wimg, wimg2 = ops.roles_wimg(w, mode, False), ops.roles_wimg(w2, mode, False)
for R in (4, 8):
    for nch in (3, 8):
        out = bufs(32)
        state = (torch.zeros(32, H, device=dev), torch.zeros(32, H, device=dev))
        edges = [round(i * T / nch) for i in range(nch + 1)]
        for k in range(nch):
            ops.roles_launch([ops.fwd_role(gx, lens, out[0], out[1], out[2], wimg, edges[k], edges[k + 1], state)], R, mode, dev)
        torch.cuda.synchronize()
        print("fwd  B 32 R %d in %d windows with carried state: bit-identical %s  status %d" % (R, nch, same_fwd(lens, ref, out), st()), flush=True)
o1, o2 = bufs(32), bufs(32)
run2 = lambda: ops.roles_launch([ops.fwd_role(gx, lens, o1[0], o1[1], o1[2], wimg), ops.fwd_role(gx2, lens2, o2[0], o2[1], o2[2], wimg2)], 8, mode, dev)
run2(); torch.cuda.synchronize()
line = "fwd  two roles (two recurrences of B 32, R 8, four XCDs each) in one launch: bit-identical %s / %s  status %d" % (same_fwd(lens, ref, o1), same_fwd(lens2, ref2, o2), st())
if not os.environ.get("CHECK_ONLY"):
    t2 = timeit(run2)
    res["fwd_two_roles_R8"] = {"us_per_step": t2[0], "median": t2[1]}
    line += "  | %.3f us/step for BOTH (median %.3f)" % t2
print(line, flush=True)
# pipeline shape: role 0 on window k while role 1 runs window k - 1 of ANOTHER sequence
nch = 8
edges = [round(i * T / nch) for i in range(nch + 1)]
o1, o2 = bufs(32), bufs(32)
s1, s2 = (torch.zeros(32, H, device=dev), torch.zeros(32, H, device=dev)), (torch.zeros(32, H, device=dev), torch.zeros(32, H, device=dev))


def pipeline():
    for k in range(nch + 1):
        roles = []
        if k < nch:
            roles.append(ops.fwd_role(gx, lens, o1[0], o1[1], o1[2], wimg, edges[k], edges[k + 1], s1))
        if k > 0:
            roles.append(ops.fwd_role(gx2, lens2, o2[0], o2[1], o2[2], wimg2, edges[k - 1], edges[k], s2))
        if len(roles) == 2:
            ops.roles_launch(roles, 8, mode, dev)
        else:
            ops.roles_launch(roles, 4, mode, dev)


pipeline(); torch.cuda.synchronize()
line = "fwd  skewed pipeline of two recurrences in %d windows (%d launches): bit-identical %s / %s  status %d" % (nch, nch + 1, same_fwd(lens, ref, o1), same_fwd(lens2, ref2, o2), st())
if not os.environ.get("CHECK_ONLY"):
    tp = timeit(pipeline)
    res["fwd_pipeline_8"] = {"us_per_step_pair": tp[0], "median": tp[1]}
    line += "  | %.3f us per step of the PAIR (median %.3f; two single launches at 4 rows: 2 x 1.62)" % tp
print(line, flush=True)

# ---------------------------------------------------------------- backward, one role
for B, R in ((32, 4), (32, 8), (64, 8), (128, 16)):
    gx, w, lens = make(B, 100 + B + R)
    sv = bufs(B); old_fwd(gx, w, lens, *sv)
    y, g, c = sv
    torch.manual_seed(5)
    dy = torch.randn(T, B, H, device=dev) * 0.1
    d0 = torch.full((T, B, 4 * H), 7.0, device=dev); old_bwd(dy, w, lens, g, c, d0)
    wimgb = ops.roles_wimg(w, mode, True)
    d1 = torch.full((T, B, 4 * H), 7.0, device=dev)
    run = lambda: ops.roles_launch([ops.bwd_role(dy, lens, g, c, d1, wimgb)], R, mode, dev, backward=True)
    run(); torch.cuda.synchronize()
    rel = float((d0 - d1).norm() / d0.norm())
    line = "bwd  B %3d R %2d: bit-identical to the launch-per-step kernel %s  rel-L2 %.2e  max %.2e  status %d" % (B, R, bool(torch.equal(d0, d1)), rel, float((d0 - d1).abs().max()), st())
    if not os.environ.get("CHECK_ONLY"):
        t_new = timeit(run); t_old = timeit(lambda: old_bwd(dy, w, lens, g, c, d0))
        res["bwd_B%d_R%d" % (B, R)] = {"us_per_step": t_new[0], "median": t_new[1], "launch_per_step_us_per_step": t_old[0], "rel_l2": rel}
        line += "  | %.3f us/step (median %.3f) against %.3f for the launch-per-step kernels" % (t_new[0], t_new[1], t_old[0])
    print(line, flush=True)
    if B == 32:
        # windows with carried state against ONE launch of the same R; image output against the fp32 rows
        for nch in (3, 8):
            d2 = torch.full((T, B, 4 * H), 7.0, device=dev)
            state = (torch.zeros(B, 4 * H, device=dev), torch.zeros(B, H, device=dev))
            edges = [round(i * T / nch) for i in range(nch + 1)]
            for k in reversed(range(nch)):
                ops.roles_launch([ops.bwd_role(dy, lens, g, c, d2, wimgb, edges[k], edges[k + 1], state, carry_in=k < nch - 1)], R, mode, dev, backward=True)
            torch.cuda.synchronize()
            print("bwd  B 32 R %d in %d windows with carried state: bit-identical to one launch %s  (max %.2e)  status %d"
                  % (R, nch, bool(torch.equal(d1, d2)), float((d1 - d2).abs().max()), st()), flush=True)
        rm = ops.row_map(lens, T, B)
        img = ops.Bf16Image.empty_rows(4 * H, rm, mode, dev)
        ops.roles_launch([ops.bwd_role(dy, lens, g, c, None, wimgb, dimg=img)], R, mode, dev, backward=True)
        img_ref = ops.Bf16Image(d1.reshape(T * B, 4 * H), colsum=True, mode=mode, rowmap=rm)
        torch.cuda.synchronize()
        nb = int((lens.sum() + B).item()) * img.ld * 2
        print("bwd  B 32 R %d image-only output: image identical to ft_bf16_image_rows(dgx) %s, column sums rel %.2e  status %d"
              % (R, bool(torch.equal(img.buf[:nb], img_ref.buf[:nb])), float((img.colsum - img_ref.colsum).norm() / img_ref.colsum.norm()), st()), flush=True)

# two backward roles
gxa, wa, la = make(32, 7); gxb, wb, lb = make(32, 8)
sa, sb = bufs(32), bufs(32); old_fwd(gxa, wa, la, *sa); old_fwd(gxb, wb, lb, *sb)
dya, dyb = torch.randn(T, 32, H, device=dev) * 0.1, torch.randn(T, 32, H, device=dev) * 0.1
wia, wib = ops.roles_wimg(wa, mode, True), ops.roles_wimg(wb, mode, True)
ra, rb = torch.empty(T, 32, 4 * H, device=dev), torch.empty(T, 32, 4 * H, device=dev)
ops.roles_launch([ops.bwd_role(dya, la, sa[1], sa[2], ra, wia)], 8, mode, dev, backward=True)
ops.roles_launch([ops.bwd_role(dyb, lb, sb[1], sb[2], rb, wib)], 8, mode, dev, backward=True)
pa, pb = torch.empty(T, 32, 4 * H, device=dev), torch.empty(T, 32, 4 * H, device=dev)
runb2 = lambda: ops.roles_launch([ops.bwd_role(dya, la, sa[1], sa[2], pa, wia), ops.bwd_role(dyb, lb, sb[1], sb[2], pb, wib)], 8, mode, dev, backward=True)
runb2(); torch.cuda.synchronize()
line = "bwd  two roles in one launch (R 8): bit-identical to one launch each %s / %s  status %d" % (bool(torch.equal(ra, pa)), bool(torch.equal(rb, pb)), st())
if not os.environ.get("CHECK_ONLY"):
    tb2 = timeit(runb2)
    res["bwd_two_roles_R8"] = {"us_per_step": tb2[0], "median": tb2[1]}
    line += "  | %.3f us/step for BOTH (median %.3f)" % tb2
print(line, flush=True)

# ---------------------------------------------------------------- phase stamps of one workgroup at R = 4 and 8 (forward and backward)
if not os.environ.get("CHECK_ONLY"):
    prof = torch.zeros(1024 * 4 * 5, dtype=torch.int64, device=dev)
    gx, w, lens = make(32, 1)
    sv = bufs(32); old_fwd(gx, w, lens, *sv)
    dy = torch.randn(T, 32, H, device=dev) * 0.1
    dd = torch.empty(T, 32, 4 * H, device=dev)
    wf, wb_ = ops.roles_wimg(w, mode, False), ops.roles_wimg(w, mode, True)
    for R in (4, 8):
        for bw in (False, True):
            prof.zero_()
            L.lib().ft_lstm_roles_debug_prof(L.ptr(prof))
            if bw:
                ops.roles_launch([ops.bwd_role(dy, lens, sv[1], sv[2], dd, wb_)], R, mode, dev, backward=True)
            else:
                o = bufs(32)
                ops.roles_launch([ops.fwd_role(gx, lens, o[0], o[1], o[2], wf)], R, mode, dev)
            torch.cuda.synchronize()
            L.lib().ft_lstm_roles_debug_prof(None)
            pr = prof.cpu().reshape(1024, 4, 5)[100:700].double()
            for wv in (0, 2):
                top, a, b, c_, npass = (pr[:, wv, k] for k in range(5))
                step = (top[1:] - top[:-1]).mean() * 10
                if bw:
                    print("stamps bwd R %d wave %d: step %.0f ns | top->gathered %.0f | ->barrier 2 passed %.0f | ->published %.0f | ->next top %.0f | poll passes %.2f"
                          % (R, wv, step, (a - top).mean() * 10, (b - a).mean() * 10, (c_ - b).mean() * 10, (top[1:] - c_[:-1]).mean() * 10, npass.mean()), flush=True)
                else:
                    print("stamps fwd R %d wave %d: step %.0f ns | sweep+mfma %.0f | reduce+barrier %.0f | epilogue->publish %.0f | ->next top %.0f | poll passes %.2f"
                          % (R, wv, step, (a - top).mean() * 10, (b - a).mean() * 10, (c_ - b).mean() * 10 if wv < 2 else 0.0,
                             (top[1:] - (c_ if wv < 2 else b)[:-1]).mean() * 10, npass.mean()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/lstm_roles_bench.json", "w"), indent=1)
