"""Tiny stand-alone launch of the dominant kernels (LSTM step fwd + bwd, bench shapes B=32, H=1024, bf16 operands) for
PMC counter passes: T steps only, so a counter pass (which serialises every dispatch) stays within seconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
T, B, H = int(os.environ.get("T", "40")), 32, 1024
dev = "cuda"
gx = torch.randn(T, B, 4 * H, device=dev) * 0.1
w = torch.randn(4 * H, H, device=dev) / H ** 0.5
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T, B, H, device=dev); gates = torch.empty(T, B, 4 * H, device=dev); cell = torch.empty(T, B, H, device=dev)
dgx = torch.empty(T, B, 4 * H, device=dev)
work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work), T, B, H, 0, 1, st), "fwd")
    L.check(L.lib().ft_lstm_seq_bwd(L.ptr(y), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work), T, B, H, 0, 1, st), "bwd")
# the two-layer wavefront chain (lstm2_fwd_step / lstm2_bwd_step): the dominant kernels of the training step
w0, wi1, w1 = (torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in range(3))
b1 = torch.zeros(4 * H, device=dev)
y0, y1 = torch.empty(T, B, H, device=dev), torch.empty(T, B, H, device=dev)
g0, g1 = torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, 4 * H, device=dev)
c0, c1 = torch.empty(T, B, H, device=dev), torch.empty(T, B, H, device=dev)
d0, d1 = torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, 4 * H, device=dev)
work2 = torch.empty(L.lib().ft_lstm2_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
for _ in range(2):
    L.check(L.lib().ft_lstm2_seq_fwd(L.ptr(gx), L.ptr(w0), L.ptr(wi1), L.ptr(b1), L.ptr(w1), L.ptr(lens), L.ptr(y0), L.ptr(g0), L.ptr(c0),
                                     L.ptr(y1), L.ptr(g1), L.ptr(c1), L.ptr(work2), T, B, H, st), "fwd2")
    L.check(L.lib().ft_lstm2_seq_bwd(L.ptr(y1), L.ptr(w0), L.ptr(wi1), L.ptr(w1), L.ptr(lens), L.ptr(g0), L.ptr(c0), L.ptr(g1), L.ptr(c1),
                                     L.ptr(d0), L.ptr(d1), L.ptr(work2), T, B, H, st), "bwd2")
torch.cuda.synchronize()
print("ok", float(y.abs().mean()), float(dgx.abs().mean()), float(y1.abs().mean()), float(d0.abs().mean()))
