"""The training step's image GEMMs at their own shapes and row maps (configs[1]: B 32, T 862, ragged lengths of the bench batch):
forward (compact output rows, 16-bit or fp32 C), input gradient (compact output rows) and weight gradient (split-K over compact
rows).  Images are built once; only the GEMM launches are timed.  TFLOP/s count the VALID rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowtron_amd import _lib as L, ops

T, B, H = 862, 32, 1024
rng = np.random.default_rng(1234)
lens_np = np.clip(rng.normal(566, 190, B), 100, T).astype(np.int64)
lens_np[0] = T
lens = torch.tensor(lens_np, dtype=torch.int32, device="cuda")
rm = ops.RowMap(lens, T, B)
rows = int(rm.rows.item())
print("valid rows + separators: %d of capacity %d" % (rows, rm.cap), flush=True)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
only = os.environ.get("GEMM_STEP_ONLY", "")
# ---- forward / dX: C[cap rows -> scattered][N] = X[cap][K] . W[N][K]^T
for name, N, K, c16 in [("gx0 fwd  K 1664 N 4096 c16", 4096, 1664, True), ("gx1 fwd  K 1024 N 4096 c16", 4096, 1024, True),
                        ("gxa fwd  K   80 N 4096 c16", 4096, 80, True),
                        ("dense    K 1024 N 1024 f32", 1024, 1024, False), ("query    K 1024 N  640 f32", 640, 1024, False),
                        ("gx0 dX   K 4096 N 1664 f32", 1664, 4096, False), ("gx1 dX   K 4096 N 1024 f32", 1024, 4096, False),
                        ("dense dX K 1024 N 1024 f32", 1024, 1024, False)]:
    if only and only not in name:
        continue
    x = torch.randn(T * B, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    xi, wi = ops.Bf16Image(x, mode=1, rowmap=rm), ops.Bf16Image(w, mode=1)
    y = torch.empty(T * B, N, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
    ms = timeit(lambda: ops.gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y, rm.cap, N, K, N, rowmap=rm, compact=1, c16=c16))
    print("%-30s %7.1f us %6.0f TF/s" % (name, ms * 1e3, 2.0 * rows * N * K / ms / 1e9), flush=True)

# ---- dW[N][K] += d[cap][N]^T . x[cap][K] (both k-major, reduction over compact rows, split-K atomics)
for name, N, K in [("dW ih0   [4096 x 1664]", 4096, 1664), ("dW hh    [4096 x 1024]", 4096, 1024), ("dW dense [1024 x 1024]", 1024, 1024),
                   ("dW query [ 640 x 1024]", 640, 1024), ("dW gxa   [4096 x   80]", 4096, 80)]:
    if only and only not in name:
        continue
    d = torch.randn(T * B, N, device="cuda")
    x = torch.randn(T * B, K, device="cuda")
    di, xi = ops.Bf16Image(d, mode=1, rowmap=rm), ops.Bf16Image(x, mode=1, rowmap=rm)
    dW = torch.zeros(N, K, device="cuda")
    ms = timeit(lambda: ops.gemm_img(di, 1, di.ptr(), xi, 1, xi.ptr(), dW, N, K, rm.cap, K, beta=1.0, splitk=True, rowmap=rm, compact=2))
    print("%-30s %7.1f us %6.0f TF/s" % (name, ms * 1e3, 2.0 * rows * N * K / ms / 1e9), flush=True)
