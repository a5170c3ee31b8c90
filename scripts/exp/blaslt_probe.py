"""What does the vendor library (hipBLASLt behind torch.mm) reach on the step's big GEMM shapes, bf16 operands, bf16 or fp32 output?
Yardstick for csrc/gemm_bf16.hip (the brief allows the library for plain GEMMs).  python scripts/exp/blaslt_probe.py"""
import torch, time
dev = "cuda"
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("fwd proj  x[M,1024] W[4096,1024]^T", 19200, 4096, 1024, "nt"), ("fwd proj  x[M,1664] W[4096,1664]^T", 19200, 4096, 1664, "nt"),
          ("dX        dY[M,4096] W[4096,1024]", 19200, 1024, 4096, "nn"), ("dense     x[M,1024] W[1024,1024]^T", 19200, 1024, 1024, "nt"),
          ("dW        dY[M,4096]^T x[M,1024]", 4096, 1024, 19200, "tn")]
for name, M, N, K, lay in shapes:
    if lay == "nt": a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16); f = lambda: a @ b.t()
    elif lay == "nn": a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16); f = lambda: a @ b
    else: a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16); f = lambda: a.t() @ b
    us = bench(f)
    line = "%-40s bf16 out: %7.1f us = %6.0f TFLOP/s" % (name, us, 2.0 * M * N * K / us / 1e6)
    try:
        if lay == "nt": g = lambda: torch.mm(a, b.t(), out_dtype=torch.float32)
        elif lay == "nn": g = lambda: torch.mm(a, b, out_dtype=torch.float32)
        else: g = lambda: torch.mm(a.t(), b, out_dtype=torch.float32)
        us32 = bench(g)
        line += " | fp32 out: %7.1f us = %6.0f TFLOP/s" % (us32, 2.0 * M * N * K / us32 / 1e6)
    except Exception as e:
        line += " | fp32 out: n/a (%s)" % type(e).__name__
    print(line, flush=True)
