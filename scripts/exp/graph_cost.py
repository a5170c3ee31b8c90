"""Host cost of replaying an LSTM chunk as a cached hipGraph vs plain launches; and overlap of 3 chains fed by ONE host thread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
T, B, H = 864, 32, 1024
dev = "cuda"
def mk():
    d = dict(gx=torch.randn(T, B, 4 * H, device=dev) * 0.1, w=torch.randn(4 * H, H, device=dev) / H ** 0.5,
             lens=torch.full((B,), T, dtype=torch.int32, device=dev), y=torch.empty(T, B, H, device=dev),
             gates=torch.empty(T, B, 4 * H, device=dev), cell=torch.empty(T, B, H, device=dev), dgx=torch.empty(T, B, 4 * H, device=dev),
             wf=torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8),
             wb=torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8))
    return d
sets = [mk() for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def fwd(s, st, s0, s1, g):
    L.check(L.lib().ft_lstm_seq_fwd_range(L.ptr(s["gx"]), L.ptr(s["w"]), L.ptr(s["lens"]), L.ptr(s["y"]), H, L.ptr(s["gates"]), L.ptr(s["cell"]),
                                          L.ptr(s["wf"]), T, B, H, 0, 1, s0, s1, g, st.cuda_stream), "fwd")
def bwd(s, st, s0, s1, g):
    L.check(L.lib().ft_lstm_seq_bwd_range(L.ptr(s["y"]), H, L.ptr(s["w"]), L.ptr(s["lens"]), L.ptr(s["gates"]), L.ptr(s["cell"]), L.ptr(s["dgx"]),
                                          L.ptr(s["wb"]), T, B, H, 0, 1, s0, s1, g, st.cuda_stream), "bwd")
for name, fn, order in (("fwd", fwd, 1), ("bwd", bwd, -1)):
    for CH in (96, 216):
        chunks = [(s0, min(T, s0 + CH)) for s0 in range(0, T, CH)][::order]
        for g in (0, 1):
            for nchain in (1, 3):
                for rep in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for (s0, s1) in chunks:                       # ONE host thread, round-robin over chains
                        for s, st in zip(sets[:nchain], streams[:nchain]):
                            fn(s, st, s0, s1, g)
                    th = time.perf_counter() - t0
                    torch.cuda.synchronize(); tt = time.perf_counter() - t0
                print("%s chunk %3d graph %d chains %d: host %.2f ms total %.2f ms" % (name, CH, g, nchain, th * 1e3, tt * 1e3), flush=True)
