"""Image GEMM (ft_gemm_img) on the training step's forward / input-gradient shapes (N(0,1) data, padded rows): 32-wide double-buffered k
stages against the 64-wide single-buffer stages (FT_GEMM_BF16_WIDE=0 | 1, read per call).  Images are built once; only the GEMM
launches are timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowtron_amd import _lib as L, ops

R = 19200
shapes = [("gx0 fwd  x[R,1664] W[4096,1664]^T", R, 4096, 1664), ("gx1 fwd  x[R,1024] W[4096,1024]^T", R, 4096, 1024),
          ("dense    x[R,1024] W[1024,1024]^T", R, 1024, 1024), ("query    x[R,1024] W[640,1024]^T", R, 640, 1024),
          ("gx0 dX   d[R,4096] Wt[1664,4096]^T", R, 1664, 4096), ("dense dX d[R,1024] Wt[1024,1024]^T", R, 1024, 1024)]
torch.manual_seed(0)
for name, M, N, K in shapes[: int(os.environ.get("GEMM_BENCH_SHAPES", "99"))]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    xi, wi = ops.Bf16Image(x, mode=1), ops.Bf16Image(w, mode=1)
    outs, res = [], []
    for wide in ("0", "1"):
        os.environ["FT_GEMM_BF16_WIDE"] = wide
        y = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y, M, N, K, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y, M, N, K, N)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((ms, 2.0 * M * N * K / ms / 1e9))
        outs.append(y)
    ref = (x.bfloat16().float() @ w.bfloat16().float().t())
    print("%-38s 32-wide %7.3f ms %5.0f TF | 64-wide %7.3f ms %5.0f TF | identical %s | max err vs torch %.2e"
          % (name, res[0][0], res[0][1], res[1][0], res[1][1], bool(torch.equal(outs[0], outs[1])), float((outs[1] - ref).abs().max())), flush=True)
os.environ.pop("FT_GEMM_BF16_WIDE", None)
