"""Fused three-recurrence forward chain (ft_lstm3_chunk_fwd) vs the separate chains: timing and bitwise equality."""
import os, sys, torch
sys.path.insert(0, ".")
from flowtron_amd import _lib as L
T, B, H = 862, 32, 1024
C = int(os.environ.get("C", "216"))
dev = "cuda"
f = dict(device=dev, dtype=torch.float32)
torch.manual_seed(0)
lens = torch.randint(100, T + 1, (B,), device=dev, dtype=torch.int32); lens[0] = T
gxa = torch.randn(T, B, 4 * H, **f) * 0.3; gx0 = torch.randn(T, B, 4 * H, **f) * 0.3
wa, w0, wi1, w1 = (torch.randn(4 * H, H, **f) / H ** 0.5 for _ in range(4))
b1 = torch.randn(4 * H, **f) * 0.1
def bufs():
    return dict(ya=torch.empty(T, B, H, **f), ga=torch.empty(T, B, 4 * H, **f), ca=torch.empty(T, B, H, **f),
                y0=torch.empty(T, B, H, **f), g0=torch.empty(T, B, 4 * H, **f), c0=torch.empty(T, B, H, **f),
                y1=torch.empty(T, B, H, **f), g1=torch.empty(T, B, 4 * H, **f), c1=torch.empty(T, B, H, **f),
                wka=torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8),
                wk2=torch.empty(L.lib().ft_lstm2_workspace_bytes(B, H), device=dev, dtype=torch.uint8))
st = torch.cuda.current_stream().cuda_stream
def separate(o):
    L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gxa), L.ptr(wa), L.ptr(lens), L.ptr(o["ya"]), H, L.ptr(o["ga"]), L.ptr(o["ca"]), L.ptr(o["wka"]), T, B, H, 0, 1, st), "a")
    L.check(L.lib().ft_lstm2_seq_fwd(L.ptr(gx0), L.ptr(w0), L.ptr(wi1), L.ptr(b1), L.ptr(w1), L.ptr(lens), L.ptr(o["y0"]), L.ptr(o["g0"]), L.ptr(o["c0"]),
                                     L.ptr(o["y1"]), L.ptr(o["g1"]), L.ptr(o["c1"]), L.ptr(o["wk2"]), T, B, H, st), "2")
def fused(o):
    def call(a0, a1, b0, b1_):
        L.check(L.lib().ft_lstm3_chunk_fwd(L.ptr(gxa), L.ptr(wa), L.ptr(o["ya"]), L.ptr(o["ga"]), L.ptr(o["ca"]), L.ptr(o["wka"]), a0, a1,
                                           L.ptr(gx0), L.ptr(w0), L.ptr(wi1), L.ptr(b1), L.ptr(w1), L.ptr(o["y0"]), L.ptr(o["g0"]), L.ptr(o["c0"]),
                                           L.ptr(o["y1"]), L.ptr(o["g1"]), L.ptr(o["c1"]), L.ptr(o["wk2"]), b0, b1_, L.ptr(lens), T, B, H, st), "3")
    edges = list(range(0, T, C)) + [T]
    n = len(edges) - 1
    for c in range(n + 1):                 # phase c: attention chunk c beside decoder chunk c-1
        a0, a1 = (edges[c], edges[c + 1]) if c < n else (0, 0)
        if c >= 1:
            b0, b1_ = edges[c - 1], edges[c] + (1 if c == n else 0)      # the last decoder chunk carries the extra launch s = T
        else:
            b0, b1_ = 0, 0
        call(a0, a1, b0, b1_)
A, Bf = bufs(), bufs()
for fn, o in ((separate, A), (fused, Bf)):
    fn(o); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn(o)
    e1.record(); torch.cuda.synchronize()
    print(fn.__name__, "%.3f ms per sequence" % (e0.elapsed_time(e1) / 3), flush=True)
for k in ("ya", "ga", "ca", "y0", "g0", "c0", "y1", "g1", "c1"):
    print(k, "equal" if torch.equal(A[k], Bf[k]) else "DIFF %.3e" % (A[k] - Bf[k]).abs().max().item())
