"""debug: where do ft_lstm_persist_*_f16 and ft_lstm_seq_*(FT_F16) differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
lib = L.lib()
T, B, H = 6, 32, 1024
torch.manual_seed(77)
gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
dy = torch.randn(T, B, H, device="cuda") * 0.1
lens_t = torch.full((B,), T, dtype=torch.int32, device="cuda")
status = torch.zeros(1, dtype=torch.int32, device="cuda")
for fmt, sfx in ((1, ""), (2, "_f16")):
    res = []
    for persist in (False, True):
        y = torch.full((T, B, H), 7.0, device="cuda")
        gates, cell = torch.zeros(T, B, 4 * H, device="cuda"), torch.zeros(T, B, H, device="cuda")
        dgx = torch.full((T, B, 4 * H), 7.0, device="cuda")
        if persist:
            work = torch.empty(lib.ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            L.check(getattr(lib, "ft_lstm_persist_fwd" + sfx)(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                                L.ptr(status), T, B, H, 1, L.stream()), "fwd")
            L.check(getattr(lib, "ft_lstm_persist_bwd" + sfx)(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work),
                                                L.ptr(status), T, B, H, 1, L.stream()), "bwd")
        else:
            work = torch.empty(lib.ft_lstm_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
            L.check(lib.ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens_t), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                        T, B, H, 0, fmt, L.stream()), "fwd")
            L.check(lib.ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens_t), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work),
                                        T, B, H, 0, fmt, L.stream()), "bwd")
        torch.cuda.synchronize()
        res.append((y, gates, cell, dgx))
    print("fmt", fmt, "status", int(status.item()))
    for name, a, b in zip(("y", "gates", "cell", "dgx"), res[0], res[1]):
        d = (a - b).abs()
        per_t = d.reshape(T, -1).max(1).values.tolist()
        print("  %-6s max diff %.3e  per t: %s  n_diff %d" % (name, d.max().item(), ["%.1e" % v for v in per_t], int((d > 0).sum())))
    if fmt == 2:
        d = (res[0][1][2] - res[1][1][2]).abs() > 0          # gates at t = 2: [B, 4H]
        print("rows with diffs:", d.any(1).sum().item(), "of", B, " cols with diffs:", d.any(0).sum().item(), "of", 4 * H)
        print("per-row counts:", d.sum(1).tolist())
        # which h_1 values are fp16-denormal (|h| < 6.1e-5) per row
        h1 = res[0][0][1]
        print("denormal-range h_1 per row:", ((h1.abs() < 6.1e-5) & (h1 != 0)).sum(1).tolist())
