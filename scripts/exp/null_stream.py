"""Does work on torch's DEFAULT stream (the HIP NULL stream) overlap with work on a torch side stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
T, B, H = 864, 32, 1024
dev = "cuda"
def mk():
    return dict(y=torch.randn(T, B, H, device=dev), w=torch.randn(4 * H, H, device=dev) / H ** 0.5, lens=torch.full((B,), T, dtype=torch.int32, device=dev),
                gates=torch.rand(T, B, 4 * H, device=dev), cell=torch.randn(T, B, H, device=dev), dgx=torch.empty(T, B, 4 * H, device=dev),
                wb=torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8))
sets = [mk() for _ in range(2)]
def bwd(s, st):
    L.check(L.lib().ft_lstm_seq_bwd_range(L.ptr(s["y"]), H, L.ptr(s["w"]), L.ptr(s["lens"]), L.ptr(s["gates"]), L.ptr(s["cell"]), L.ptr(s["dgx"]),
                                          L.ptr(s["wb"]), T, B, H, 0, 1, 0, T, 1, st.cuda_stream), "bwd")
default = torch.cuda.default_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("default stream handle:", default.cuda_stream, "side:", s1.cuda_stream, s2.cuda_stream)
for name, pair in (("side+side", (s1, s2)), ("default+side", (default, s1))):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bwd(sets[0], pair[0]); bwd(sets[1], pair[1])
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); bwd(sets[0], pair[0]); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("%s: two chains %.2f ms (one chain alone %.2f ms)" % (name, tt * 1e3, t1 * 1e3), flush=True)
# GEMM-like heavy kernel on a side stream concurrently with a chain on the default stream
a = torch.randn(8192, 8192, device=dev); bmat = torch.randn(8192, 8192, device=dev)
for name, (sc, sg) in (("chain default, matmul side", (default, s1)), ("chain side, matmul side", (s2, s1))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bwd(sets[0], sc)
        with torch.cuda.stream(sg):
            for _ in range(10): c = a @ bmat
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(sg):
        for _ in range(10): c = a @ bmat
    torch.cuda.synchronize(); tm = time.perf_counter() - t0
    print("%s: together %.2f ms (matmuls alone %.2f ms)" % (name, tt * 1e3, tm * 1e3), flush=True)
