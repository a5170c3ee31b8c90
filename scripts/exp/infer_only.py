import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, flowtron
torch.manual_seed(1234)
model = flowtron.Flowtron(**bench.MODEL_CONFIG); bench.init_weights(model, 1234); model = model.cuda().eval()
z = torch.randn(1, 80, 400, device="cuda") * 0.5
text = torch.randint(0, 185, (1, 69), device="cuda"); spk = torch.zeros(1, dtype=torch.long, device="cuda")
for g in ("1", "0"):
    os.environ["FLOWTRON_DECODE_GRAPH"] = g
    model.infer(z, spk, text, gate_threshold=1.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); mel, _ = model.infer(z, spk, text, gate_threshold=1.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("graph", g, "frames", mel.shape[2], "ms", dt * 1e3, "rtf", dt / (mel.shape[2] * 256 / 22050), flush=True)
