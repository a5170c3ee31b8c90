"""persistent BiLSTM vs the pair chain vs the fp32 oracle (per direction), and timing.  GPU box: python scripts/exp/bilstm_persist_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
from oracle import flowtron_oracle as O
T, B, H, fmt = 157, 32, 256, 1
torch.manual_seed(1)
lens = [T] + [max(1, T - 4 * i - (i % 3)) for i in range(1, B)]
lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
gx = [torch.randn(T, B, 4 * H, device="cuda") * 0.7 for _ in range(2)]
w = [torch.randn(4 * H, H, device="cuda") / H ** 0.5 for _ in range(2)]
dy = torch.randn(T, B, 2 * H, device="cuda") * 0.1
f = dict(device="cuda", dtype=torch.float32)
lib = L.lib()


def run(persistent, bufs=None):
    y = torch.full((T, B, 2 * H), 7.0, **f)
    gates = [torch.zeros(T, B, 4 * H, **f) for _ in range(2)]
    cell = [torch.zeros(T, B, H, **f) for _ in range(2)]
    dgx = [torch.full((T, B, 4 * H), 7.0, **f) for _ in range(2)]
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    if persistent:
        wk = torch.empty(lib.ft_bilstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
        fw = lambda: L.check(lib.ft_bilstm_persist_fwd(L.ptr(gx[0]), L.ptr(gx[1]), L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(y), 2 * H, L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(wk), L.ptr(status), T, B, H, L.stream()), "pf")
        bw = lambda: L.check(lib.ft_bilstm_persist_bwd(L.ptr(dy), 2 * H, L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(dgx[0]), L.ptr(dgx[1]), L.ptr(wk), L.ptr(status), T, B, H, L.stream()), "pb")
    else:
        wk = [torch.empty(lib.ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8) for _ in range(2)]
        fw = lambda: L.check(lib.ft_lstm_bidir_seq_fwd(L.ptr(gx[0]), L.ptr(gx[1]), L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(y), 2 * H, L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(wk[0]), L.ptr(wk[1]), T, B, H, L.stream()), "cf")
        bw = lambda: L.check(lib.ft_lstm_bidir_seq_bwd(L.ptr(dy), 2 * H, L.ptr(w[0]), L.ptr(w[1]), L.ptr(lens_t), L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]), L.ptr(dgx[0]), L.ptr(dgx[1]), L.ptr(wk[0]), L.ptr(wk[1]), T, B, H, L.stream()), "cb")
    fw(); bw(); torch.cuda.synchronize()
    ts = []
    for fn in (fw, bw):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        ts.append(best)
    return y, gates, cell, dgx, int(status.item()), ts


res = [run(False), run(True)]
# fp32 oracle per direction (explicit recurrence; zero input projection: gx is given -> emulate with identity input)
ref_y, ref_d = [], []
lens_c = torch.tensor(lens)
for d in range(2):
    g = gx[d].cpu().requires_grad_(True)
    eye = torch.eye(4 * H)
    yy = O.lstm_cell_seq(g, lens_c, eye, w[d].cpu(), torch.zeros(4 * H), torch.zeros(4 * H), reverse=bool(d))
    (yy * dy.cpu()[:, :, d * H:(d + 1) * H]).sum().backward()
    ref_y.append(yy.detach()); ref_d.append(g.grad)
for name, (y, gates, cell, dgx, st, ts) in zip(("chain", "persistent"), res):
    print("%-10s status %d  fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)" % (name, st, ts[0], ts[0] / T, ts[1], ts[1] / T))
    for d in range(2):
        ey = float((y[:, :, d * H:(d + 1) * H].cpu() - ref_y[d]).abs().max())
        ed = float((dgx[d].cpu() - ref_d[d]).norm() / ref_d[d].norm())
        print("   dir %d: |y - oracle| max %.2e   dgx rel-L2 vs oracle %.2e" % (d, ey, ed))
for d in range(2):
    a, b = res[0], res[1]
    print("dir %d chain vs persistent: y %.2e  gates %.2e  cell %.2e  dgx %.2e (max |dgx| %.2e)" % (
        d, float((a[0][:, :, d * H:(d + 1) * H] - b[0][:, :, d * H:(d + 1) * H]).abs().max()), float((a[1][d] - b[1][d]).abs().max()),
        float((a[2][d] - b[2][d]).abs().max()), float((a[3][d] - b[3][d]).abs().max()), float(a[3][d].abs().max())))
