"""Experiment: do independent latency-bound LSTM launch chains overlap when issued on separate HIP streams?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L

T, B, H = 862, 32, 1024
dev = "cuda"
def mk():
    gx = torch.randn(T, B, 4 * H, device=dev) * 0.1
    w = torch.randn(4 * H, H, device=dev) / H ** 0.5
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    y = torch.empty(T, B, H, device=dev); gates = torch.empty(T, B, 4 * H, device=dev); cell = torch.empty(T, B, H, device=dev)
    work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
    dgx = torch.empty(T, B, 4 * H, device=dev)
    return dict(gx=gx, w=w, lens=lens, y=y, gates=gates, cell=cell, work=work, dgx=dgx)
sets = [mk() for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def fwd(s, st):
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(s["gx"]), L.ptr(s["w"]), L.ptr(s["lens"]), L.ptr(s["y"]), H, L.ptr(s["gates"]), L.ptr(s["cell"]),
                                    L.ptr(s["work"]), T, B, H, 0, 1, st.cuda_stream), "fwd")
def bwd(s, st):
    L.check(L.lib().ft_lstm_seq_bwd(L.ptr(s["y"]), H, L.ptr(s["w"]), L.ptr(s["lens"]), L.ptr(s["gates"]), L.ptr(s["cell"]), L.ptr(s["dgx"]),
                                    L.ptr(s["work"]), T, B, H, 0, 1, st.cuda_stream), "bwd")
for fn, name in ((fwd, "fwd"), (bwd, "bwd")):
    for s, st in zip(sets, streams):
        fn(s, st)
    torch.cuda.synchronize()
    for n in (1, 2, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, st in zip(sets[:n], streams[:n]):
            fn(s, st)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%s: %d concurrent chains of %d steps: host enqueue %.2f ms, total %.2f ms (%.2f us/step/chain)" % (name, n, T, t_host * 1e3, t_all * 1e3, t_all * 1e6 / T), flush=True)
    # interleaved enqueue (round-robin per-step is impossible through the seq API; approximate with chunks via threads)
import threading
for fn, name in ((fwd, "fwd"), (bwd, "bwd")):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=fn, args=(s, st)) for s, st in zip(sets, streams)]
    [t.start() for t in th]; [t.join() for t in th]
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%s: 3 chains enqueued from 3 host threads: host %.2f ms total %.2f ms" % (name, t_host * 1e3, (time.perf_counter() - t0) * 1e3), flush=True)
