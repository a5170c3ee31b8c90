"""Why does the BACKWARD recurrence at 8 rows per XCD group slow down when a second role runs beside it (2.19 -> 2.55 us per step), where
the forward does not?  One role / two roles, fp32 dgx rows against the 16-bit image only, both roles on the same or on different data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flowtron_amd import _lib as L
from flowtron_amd import ops

T, H, B = 862, 1024, 32
dev = torch.device("cuda", 0)
mode = L.FT_BF16


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / T)
    return sorted(ts)[len(ts) // 2]


def make(seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(int(T * 0.45), T + 1, (B,), generator=g, dtype=torch.int32); lens[0] = T
    lens = lens.to(dev)
    gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
    w = torch.randn(4 * H, H, device=dev) / 32
    y, gt, c = torch.empty(T, B, H, device=dev), torch.empty(T, B, 4 * H, device=dev), torch.empty(T, B, H, device=dev)
    ops.roles_launch([ops.fwd_role(gx, lens, y, gt, c, ops.roles_wimg(w, mode, False))], 4, mode, dev)
    dy = torch.randn(T, B, H, device=dev) * 0.1
    return dict(lens=lens, g=gt, c=c, dy=dy, w=ops.roles_wimg(w, mode, True), dgx=torch.empty(T, B, 4 * H, device=dev),
                img=ops.Bf16Image.empty_rows(4 * H, ops.row_map(lens, T, B), mode, dev))


a, b = make(1), make(2)
role = lambda d, image: ops.bwd_role(d["dy"], d["lens"], d["g"], d["c"], None if image else d["dgx"], d["w"], dimg=d["img"] if image else None)
for image in (False, True):
    tag = "16-bit image only" if image else "fp32 dgx rows     "
    for R in (4, 8):
        print("%s  one role   R %d: %.3f us per step" % (tag, R, timeit(lambda: ops.roles_launch([role(a, image)], R, mode, dev, backward=True))), flush=True)
    print("%s  two roles  R 8: %.3f us per step for both" % (tag, timeit(lambda: ops.roles_launch([role(a, image), role(b, image)], 8, mode, dev, backward=True))), flush=True)
ops.check_persist_status()
