"""lstm_persist_bwd_k at the bench shape: fp32 dgx only | fp32 dgx + compact 16-bit image | image only (us per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L, ops
import bench
T, B, H = 862, 32, 1024
torch.manual_seed(0)
lens = bench.synth_batch(32, 1234 + 7)["out_lens"].int().cuda()
gx = torch.randn(T, B, 4 * H, device="cuda") * 0.5
w = torch.randn(4 * H, H, device="cuda") / H ** 0.5
dy = torch.randn(T, B, H, device="cuda") * 0.1
y, g, c, d = torch.empty(T, B, H, device="cuda"), torch.empty(T, B, 4 * H, device="cuda"), torch.empty(T, B, H, device="cuda"), torch.empty(T, B, 4 * H, device="cuda")
st = torch.zeros(1, dtype=torch.int32, device="cuda")
wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device="cuda", dtype=torch.uint8)
L.check(L.lib().ft_lstm_persist_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(g), L.ptr(c), L.ptr(wp), L.ptr(st), T, B, H, 1, L.stream()), "f")
rm = ops.RowMap(lens, T, B)
img = ops.Bf16Image.empty_rows(4 * H, rm, 1, torch.device("cuda"))
rows_alloc = img.buf.numel() // (2 * img.ld)


def run(kind):
    a = [L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(g), L.ptr(c), L.ptr(d) if kind != "img" else None, L.ptr(wp), L.ptr(st), T, B, H, 11]
    if kind == "f32":
        L.check(L.lib().ft_lstm_persist_bwd(*a, L.stream()), "b")
    else:
        L.check(L.lib().ft_lstm_persist_bwd_img(*a, L.ptr(img.buf), img.ld, rows_alloc, L.ptr(img.colsum), L.stream()), "bi")


for rnd in range(3):
    for kind in ("f32", "both", "img"):
        run(kind); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(kind); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / T)
        print("round %d  %-5s min %.3f us/step  status %d" % (rnd, kind, min(ts), int(st.item())), flush=True)
