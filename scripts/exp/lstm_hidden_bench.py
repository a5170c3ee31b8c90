"""One LSTM layer (ops.lstm_layer: input projection + recurrence, forward + backward) at hidden sizes below the persistent kernels' 1024:
the zero-padded 1024-unit twin on the persistent kernels (ops._PAD_H, default) against the launch-per-step kernels.  T 862, B 32, the
bench batch's ragged lengths, bf16 operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowtron_amd import _lib as L, ops

T, B, K = 862, 32, 1024
rng = np.random.default_rng(1234)
lens_np = np.clip(rng.normal(566, 190, B), 100, T).astype(np.int64); lens_np[0] = T
lens = torch.tensor(lens_np, dtype=torch.int32, device="cuda")
torch.manual_seed(0)
for H in (256, 512, 768, 1024):
    x = torch.randn(T, B, K, device="cuda")
    p = [torch.randn(4 * H, K, device="cuda") * 0.03, torch.randn(4 * H, H, device="cuda") / H ** 0.5, torch.zeros(4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")]
    go = torch.randn(T, B, H, device="cuda")
    line = "H %4d:" % H
    for pad in (True, False):
        ops._PAD_H = pad
        def step():
            d = [t.detach().requires_grad_(True) for t in [x] + p]
            y = ops.lstm_layer(d[0], lens, d[1], d[2], d[3], d[4], mode=1, rowmap=ops.RowMap(lens, T, B), fill="dx")
            y.backward(go)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record(); torch.cuda.synchronize()
        line += "  %s %7.2f ms" % ("persistent (padded to 1024)" if pad and H < 1024 else "persistent" if H == 1024 else "launch per step", e0.elapsed_time(e1) / 5)
        if H == 1024:
            break
    print(line, flush=True)
ops.check_persist_status()
