"""Attention forward / backward at the bench shape (T 862, B 32, L 157, A 640, the bench's own lengths, prior on): ms per call of
ft_attention_fwd / ft_attention_bwd, HIP events.  Run on the GPU box: python scripts/exp/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from flowtron_amd import _lib as L, ops

bb = bench.synth_batch(32, 1234 + 7)
T, Lk, B, A = bb["mel"].shape[2], bb["text"].shape[1], 32, 640
in_lens, out_lens = bb["in_lens"].int().cuda(), bb["out_lens"].int().cuda()
torch.manual_seed(0)
Q = torch.randn(T, B, A, device="cuda") * 0.7
valid_t = (torch.arange(T, device="cuda")[:, None] < out_lens[None, :])[..., None]
Q = torch.where(valid_t, Q, Q[-1:, :1, :].expand_as(Q) * 0 + 0.123)      # padded frames: one repeated query row, like the model's
K = torch.randn(Lk, B, A, device="cuda") * 0.7
v = torch.randn(A, device="cuda") * 0.3
prior = bench.beta_binomial_prior_batch(bb["in_lens"], bb["out_lens"], T, Lk).cuda()
Qr, Kr, vr = Q.clone().requires_grad_(True), K.clone().requires_grad_(True), v.clone().reshape(1, -1).requires_grad_(True)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


attn, lp = ops.AttentionScoresFn.apply(Qr, Kr, vr, in_lens, prior, 1.0)
g1, g2 = torch.randn_like(attn) * valid_t.permute(1, 0, 2).float(), torch.randn_like(lp) * 0.1 * valid_t.permute(1, 0, 2).float()
fwd = timeit(lambda: ops.AttentionScoresFn.apply(Q, K, v.reshape(1, -1), in_lens, prior, 1.0))


def bwd():
    a, l_ = ops.AttentionScoresFn.apply(Qr, Kr, vr, in_lens, prior, 1.0)
    torch.autograd.grad([a, l_], [Qr, Kr, vr], [g1, g2])


both = timeit(bwd)
print("attention at T %d B %d L %d A %d: forward %.3f ms, forward + backward %.3f ms (backward %.3f)" % (T, B, Lk, A, fwd, both, both - fwd))
print("checksums:", float(attn.sum()), float(lp[torch.isfinite(lp)].abs().mean()))
