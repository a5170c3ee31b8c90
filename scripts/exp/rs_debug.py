"""debug: reduce-scatter backward recurrence (transport 21) vs the launch-per-step kernel, error per time step / row / unit block"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
T, B, H = int(os.environ.get("T", 6)), int(os.environ.get("B", 32)), 1024
dev = "cuda"
torch.manual_seed(0)
gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
w = torch.randn(4 * H, H, device=dev) / H ** 0.5
dy = torch.randn(T, B, H, device=dev) * 0.1
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
if os.environ.get("RAGGED"):
    lens = torch.tensor([max(1, T - (i % 5)) for i in range(B)], dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
y, g, c = torch.empty(T, B, H, device=dev), torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, H, device=dev)
ws = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(g), L.ptr(c), L.ptr(ws), T, B, H, 0, 1, L.stream()), "fwd")
d0 = torch.full((T, B, 4 * H), 7.0, device=dev)
L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g), L.ptr(c), L.ptr(d0), L.ptr(ws), T, B, H, 0, 1, L.stream()), "bwd step")
for ng in (11, 21):
    d1 = torch.full((T, B, 4 * H), 7.0, device=dev)
    L.check(L.lib().ft_lstm_persist_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g), L.ptr(c), L.ptr(d1), L.ptr(wp), L.ptr(status), T, B, H, ng, L.stream()), "bwd persist")
    torch.cuda.synchronize()
    print("ng", ng, "status", int(status.item()), "equal", bool(torch.equal(d0, d1)), "rel", float((d0 - d1).norm() / d0.norm()))
    if ng == 21:
        for t in range(T - 1, -1, -1):
            e = (d0[t] - d1[t]).abs()
            print(" t=%d  max err %.3e  ref max %.3e | per row-group of 4: %s" % (t, float(e.max()), float(d0[t].abs().max()),
                  " ".join("%.1e" % float(e[4 * k:4 * k + 4].max()) for k in range(min(8, (B + 3) // 4)))))
        t = max(T - 2, 0)
        e = (d0[t] - d1[t]).abs().reshape(B, 4, 32, 32)       # [b][gate][cu][unit]
        print(" t=%d error by unit-block (cu) of row 0, gate 0:" % t, " ".join("%.0e" % float(e[0, 0, k].max()) for k in range(32)))
        print(" t=%d error by row (0..7), all gates/units:" % t, " ".join("%.1e" % float(e[b].max()) for b in range(min(8, B))))
        # the dh that would explain d1: compare the recurrent product directly
        da_next = d0[t + 1] if t + 1 < T else None
        if da_next is not None:
            dh_ref = da_next.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float()        # [B, H]
            print(" |dh_rec| of step t+1 -> t: max %.3e" % float(dh_ref.abs().max()))
