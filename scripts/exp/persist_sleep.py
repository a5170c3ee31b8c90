"""forward persistent recurrence: us/step as a function of FT_PERSIST_SLEEP (s_sleep units before a step's first poll)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
lib = L.lib()
T, B, H = 862, 32, 1024
torch.manual_seed(0)
gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
y, gates, cell = torch.empty(T, B, H, device="cuda"), torch.empty(T, B, 4 * H, device="cuda"), torch.empty(T, B, H, device="cuda")
work = torch.empty(lib.ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
status = torch.zeros(1, dtype=torch.int32, device="cuda")
def run():
    L.check(lib.ft_lstm_persist_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work), L.ptr(status), T, B, H, 1, L.stream()), "fwd")
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / T)
print("FT_PERSIST_SLEEP=%s: min %.3f median %.3f us/step status %d" % (os.environ.get("FT_PERSIST_SLEEP", "0"), min(ts), sorted(ts)[2], int(status.item())))
