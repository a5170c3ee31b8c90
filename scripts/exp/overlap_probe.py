"""Can a weight-gradient GEMM on a second stream hide under the attention backward (VALU / transcendental-bound, matrix cores idle)?
attention fwd+bwd alone, the split-K dW GEMM [4096 x 1024] over the compact rows alone (x 3), and both issued on two streams."""
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
K = torch.randn(Lk, B, A, device="cuda") * 0.7
v = torch.randn(A, device="cuda") * 0.3
prior = bench.beta_binomial_prior_batch(bb["in_lens"], bb["out_lens"], T, Lk).cuda()
Qr, Kr, vr = Q.clone().requires_grad_(True), K.clone().requires_grad_(True), v.clone().reshape(1, -1).requires_grad_(True)
attn, lp = ops.AttentionScoresFn.apply(Qr, Kr, vr, in_lens, prior, 1.0)
g1, g2 = torch.randn_like(attn), torch.randn_like(lp) * 0.1

rm = ops.RowMap(out_lens, T, B)
d = torch.randn(T * B, 4096, device="cuda") * 0.01
x = torch.randn(T * B, 1024, device="cuda")
di, xi = ops.Bf16Image(d, mode=1, rowmap=rm), ops.Bf16Image(x, mode=1, rowmap=rm)
dW = torch.zeros(4096, 1024, device="cuda")
side = torch.cuda.Stream()


def attn_bwd():
    torch.autograd.grad([attn, lp], [Qr, Kr, vr], [g1, g2], retain_graph=True)


def gemms():
    for _ in range(3):
        ops.gemm_img(di, 1, di.ptr(), xi, 1, xi.ptr(), dW, 4096, 1024, rm.cap, 1024, beta=1.0, splitk=True, rowmap=rm, compact=2)


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gemms()
    attn_bwd()
    torch.cuda.current_stream().wait_stream(side)


def timeit(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


ta, tg, tb = timeit(attn_bwd), timeit(gemms), timeit(both)
print("attention backward %.3f ms | 3 dW GEMMs %.3f ms | sum %.3f | on two streams %.3f ms" % (ta, tg, ta + tg, tb))
