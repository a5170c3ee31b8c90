"""Old (fp32-staging) vs new (bf16 image + DMA) GEMM path on the training step's shapes.  Prints TFLOP/s incl. the image passes."""
import math, sys, torch
sys.path.insert(0, ".")
from flowtron_amd import _lib as L, ops

R = 27584
shapes = [  # name, M, N, K, layout (ta, tb), splitk
    ("gx0 fwd   x[R,1664] W[4096,1664]^T", R, 4096, 1664, (False, True), False),
    ("gx0 dX    d[R,4096] W[4096,1664]", R, 1664, 4096, (False, False), False),
    ("gx0 dW    d[R,4096]^T x[R,1664]", 4096, 1664, R, (True, False), True),
    ("hh  dW    d[R,4096]^T h[R,1024]", 4096, 1024, R, (True, False), True),
    ("dense fwd x[R,1024] W[1024,1024]^T", R, 1024, 1024, (False, True), False),
    ("dense dW  d[R,1024]^T x[R,1024]", 1024, 1024, R, (True, False), True),
    ("attgx fwd x[R,80] W[4096,80]^T", R, 4096, 80, (False, True), False),
    ("attgx dX  d[R,4096] W[4096,80]", R, 80, 4096, (False, False), False),
    ("query fwd x[R,1024] W[640,1024]^T", R, 640, 1024, (False, True), False),
]
for name, M, N, K, (ta, tb), sk in shapes:
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    Cm = torch.empty(M, N, device="cuda")
    sAm, sAk = (1, M) if ta else (K, 1)
    sBk, sBn = (1, K) if tb else (N, 1)
    res = []
    for images in (False, True):
        ops._BF16_IMAGES = images
        for _ in range(2):
            ops.gemm_raw(A, B, Cm, M, N, K, sAm, sAk, sBk, sBn, N, mode=1, splitk=sk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_raw(A, B, Cm, M, N, K, sAm, sAk, sBk, sBn, N, mode=1, splitk=sk)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res.append((ms, 2.0 * M * N * K / ms / 1e9))
    print(f"{name:40s} staging {res[0][0]:7.3f} ms {res[0][1]:6.0f} TF | images {res[1][0]:7.3f} ms {res[1][1]:6.0f} TF", flush=True)
