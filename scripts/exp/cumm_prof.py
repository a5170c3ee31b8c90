"""Stage stamps of the fused cumulative-attention frame kernels (csrc/cumm_fused.hip) at the bench shape: where a frame goes.
usage (GPU box): python scripts/exp/cumm_prof.py [T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowtron_amd import _lib as L          # noqa: E402
from flowtron_amd import ops                 # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, Lk, E, A = 32, 157, 640, 640
torch.manual_seed(0)
f = dict(device="cuda", dtype=torch.float32)
lens = torch.randint(60, Lk + 1, (B,), dtype=torch.int32)
lens[0] = Lk
lens = lens.sort(descending=True).values.cuda()
Q, V, text = torch.randn(T, B, A, **f) * 0.7, torch.randn(Lk, B, A, **f), torch.randn(Lk, B, E, **f) * 0.7
wk, v = torch.randn(A, E, **f) / E ** 0.5, torch.randn(1, A, **f) / A ** 0.5 * 4
w1, b1, w2, b2 = torch.randn(32, 2, 5, **f) * 0.5, torch.randn(32, **f) * 0.2, torch.randn(E, 32, 3, **f) * 0.2, torch.randn(E, **f) * 0.2
leaves = [t.requires_grad_(True) for t in (Q, V, text, wk, v, w1, b1, w2, b2)]
prof = torch.zeros(2, 4096, 16, device="cuda", dtype=torch.int64)
for rep in range(2):
    if rep == 1:
        L.check(L.lib().ft_cumm_attn_debug_prof(L.ptr(prof)), "prof")
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    c, a_, lp = ops.CummAttnSeqFn.apply(*leaves, lens, 1.0, L.FT_BF16)
    e1.record()
    (c.sum() + (a_ * torch.randn_like(a_)).sum()).backward()
    e2.record()
    torch.cuda.synchronize()
    print("rep %d: forward %.3f ms (%.1f us/frame), backward %.3f ms (%.1f us/frame)" % (
        rep, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T, e1.elapsed_time(e2), e1.elapsed_time(e2) * 1e3 / T))
L.lib().ft_cumm_attn_debug_prof(None)
p = prof.cpu().double() * 0.01                    # us
for d, name, n, labels in ((0, "forward", 6, ["top->softmax done", "->h1", "->km tile", "->K GEMM", "->scores out"]),
                           (1, "backward", 7, ["top->h1", "->dK tile", "->dkm GEMM", "->dpre2 tile", "->dcol2", "->end"])):
    fr = p[d, 5:T - 5, :n]
    dt = (fr[:, 1:] - fr[:, :-1]).mean(0)
    gap = (p[d, 6:T - 5, 0] - p[d, 5:T - 6, n - 1]).abs().mean() if d == 0 else (p[d, 5:T - 6, 0] - p[d, 6:T - 5, n - 1]).abs().mean()
    if d == 1:
        print("backward extra stamps (us after the previous main stamp): col2 stream written +%.2f (after h1), streams copied +%.2f (after dpre2 tile)" % (
            (p[1, 5:T - 5, 8] - p[1, 5:T - 5, 1]).mean().item(), (p[1, 5:T - 5, 7] - p[1, 5:T - 5, 4]).mean().item()))
    print(name, "per-stage us:", ", ".join("%s %.2f" % (lb, x) for lb, x in zip(labels, dt.tolist())), "| in-kernel %.2f" % (fr[:, -1] - fr[:, 0]).mean().item(),
          "| end->next top %.2f" % gap.item())
