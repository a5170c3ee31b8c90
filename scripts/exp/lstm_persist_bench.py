"""Persistent (one launch per sequence) vs launch-per-step attention-LSTM forward: B = 32, H = 1024, T = 862 (bench shape).
Prints us/step for each variant + bit-equality of the outputs.  Run on the GPU box: python scripts/exp/lstm_persist_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L

T, B, H = int(os.environ.get("T", 862)), 32, 1024
dev = "cuda"
torch.manual_seed(0)
gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
w = torch.randn(4 * H, H, device=dev) / H ** 0.5
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
lens[5:] -= torch.arange(B - 5, dtype=torch.int32, device=dev) * 7
status = torch.zeros(1, dtype=torch.int32, device=dev)


def bufs():
    return (torch.empty(T, B, H, device=dev), torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, H, device=dev))


def run_step(y, g, c, work):
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(g), L.ptr(c), L.ptr(work), T, B, H, 0, 1, L.stream()), "step")


def run_persist(ng, y, g, c, work):
    L.check(L.lib().ft_lstm_persist_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(g), L.ptr(c), L.ptr(work), L.ptr(status),
                                        T, B, H, ng, L.stream()), "persist")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    return min(ts), sorted(ts)[len(ts) // 2]


res = {}
y0, g0, c0 = bufs()
w0 = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
res["step"] = timeit(lambda: run_step(y0, g0, c0, w0))
print("launch-per-step: min %.3f median %.3f us/step" % res["step"], flush=True)
wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
for ng in [int(v) for v in os.environ.get("FWD_NGS", "1").split(",")]:
    y1, g1, c1 = bufs()
    y1.fill_(7.0)
    r = timeit(lambda: run_persist(ng, y1, g1, c1, wp))
    st = int(status.item())
    act = (torch.arange(T, device=dev)[:, None] < lens[None, :])
    eq = bool(torch.equal(y0, y1)) and bool(torch.equal(g0[act], g1[act])) and bool(torch.equal(c0[act], c1[act]))
    res["persist%d" % ng] = r + (st, eq, float((y0 - y1).abs().max()))
    print("persistent ng=%d (1 = XCD-local nt loads, 9 = XCD-local sc1 loads): min %.3f median %.3f us/step  status %d  bit-identical %s  max|dy| %.3e" % ((ng,) + res["persist%d" % ng]), flush=True)
    status.zero_()
# ---- phase stamps of one workgroup (forward, XCD-local transport)
prof = torch.zeros(1024 * 4 * 5, dtype=torch.int64, device=dev)
L.lib().ft_lstm_persist_debug_prof(L.ptr(prof))
yp, gp_, cp = bufs()
run_persist(int(os.environ.get("PROF_NG", "11")), yp, gp_, cp, wp)
torch.cuda.synchronize()
L.lib().ft_lstm_persist_debug_prof(None)
pr = prof.cpu().reshape(1024, 4, 5)[100:800].double()
print("stamps of transport", os.environ.get("PROF_NG", "11"))
for wv in range(4):
    top, swp, bar, pub, npass = (pr[:, wv, k] for k in range(5))
    step = (top[1:] - top[:-1]).mean() * 10
    if os.environ.get("PROF_NG", "11") == "31":      # M-split kernel: top -> gathered + LDS written | barrier + LDS reads + MFMAs | cell update -> published | -> next top
        print("wave %d (ng 31): step %.0f ns | top->gathered %.0f | stores + barrier %.0f | LDS reads + mfma %.0f | epilogue->publish %.0f | publish->next top %.0f"
              % (wv, step, (swp - top).mean() * 10, (npass - swp).mean() * 10, (bar - npass).mean() * 10, (pub - bar).mean() * 10, (top[1:] - pub[:-1]).mean() * 10), flush=True)
        continue
    print("wave %d: step %.0f ns | sweep+mfma %.0f | reduce+barrier %.0f | epilogue->publish %.0f | publish->next top %.0f | poll passes %.2f"
          % (wv, step, ((swp - top).mean()) * 10, ((bar - swp).mean()) * 10, ((pub - bar).mean()) * 10 if wv < 2 else 0.0,
             ((top[1:] - (pub if wv < 2 else bar)[:-1]).mean()) * 10, npass.mean()), flush=True)
# ---- backward
dy = torch.randn(T, B, H, device=dev) * 0.1
act3 = (torch.arange(T, device=dev)[:, None] < lens[None, :])
d0 = torch.empty(T, B, 4 * H, device=dev)


def run_step_bwd(dgx, work):
    L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g0), L.ptr(c0), L.ptr(dgx), L.ptr(work), T, B, H, 0, 1, L.stream()), "bstep")


def run_persist_bwd(ng, dgx, work):
    L.check(L.lib().ft_lstm_persist_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g0), L.ptr(c0), L.ptr(dgx), L.ptr(work), L.ptr(status),
                                        T, B, H, ng, L.stream()), "bpersist")


res["bwd_step"] = timeit(lambda: run_step_bwd(d0, w0))
print("backward launch-per-step: min %.3f median %.3f us/step" % res["bwd_step"], flush=True)
for ng in [int(v) for v in os.environ.get("BWD_NGS", "1,11,21").split(",")]:
    d1 = torch.full((T, B, 4 * H), 7.0, device=dev)
    r = timeit(lambda: run_persist_bwd(ng, d1, wp))
    st = int(status.item())
    rel = float((d0 - d1).norm() / d0.norm())
    res["bwd_persist%d" % ng] = r + (st, bool(torch.equal(d0, d1)), float((d0 - d1).abs().max()), rel)
    print("backward persistent ng=%d: min %.3f median %.3f us/step  status %d  bit-identical %s  max|d| %.3e  rel-L2 %.3e" % ((ng,) + res["bwd_persist%d" % ng]), flush=True)
    status.zero_()
# ---- phase stamps of one workgroup (backward, XCD-local transport)
prof.zero_()
L.lib().ft_lstm_persist_debug_prof(L.ptr(prof))
run_persist_bwd(int(os.environ.get("PROF_NG_BWD", "21")), d1, wp)
torch.cuda.synchronize()
L.lib().ft_lstm_persist_debug_prof(None)
pr = prof.cpu().reshape(1024, 4, 5)[100:800].double()
for wv in range(4):
    top, swp, bar, pub, npass = (pr[:, wv, k] for k in range(5))
    step = (top[1:] - top[:-1]).mean() * 10
    print("bwd wave %d (ng 21: top->gathered | gathered->barrier 2 passed | barrier 2->published | published->next top): step %.0f ns | %.0f | %.0f | %.0f | %.0f | poll passes %.2f"
          % (wv, step, ((swp - top).mean()) * 10, ((bar - swp).mean()) * 10, ((pub - bar).mean()) * 10,
             ((top[1:] - pub[:-1]).mean()) * 10, npass.mean()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/lstm_persist_bench.json", "w"))
