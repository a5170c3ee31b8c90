"""Stage stamps of the persistent decode (workgroup 0): where a frame's time goes.  GPU box: python scripts/exp/decode_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FLOWTRON_MFMA"] = os.environ.get("DEC_MODE", "bf16")
import torch
import flowtron, bench
from flowtron_amd import _lib as L
torch.manual_seed(0)
m = flowtron.Flowtron(**bench.MODEL_CONFIG).cuda().eval()
bench.init_weights(m, 1)
z = torch.randn(1, 80, 300, device="cuda") * 0.5
text = torch.randint(0, 185, (1, 69), device="cuda")
spk = torch.zeros(1, dtype=torch.long, device="cuda")
m.infer(z, spk, text, gate_threshold=1.0)
prof = torch.zeros(512 * 12, dtype=torch.int64, device="cuda")
L.lib().ft_decode_debug_prof(L.ptr(prof))
m.infer(z, spk, text, gate_threshold=1.0)
torch.cuda.synchronize()
L.lib().ft_decode_debug_prof(None)
p = prof.cpu().reshape(512, 12)[20:280].double()
names = ["top->o", "o->S1 done(publish hatt)", "S2 gather hatt", "S3a gather q", "S3b gather scores", "S4 gather ctx", "S5 gather h0", "S6 gather h1",
         "S7 gather u1", "S8 gather u2", "->next top"]
seq = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
for a, b, nm in zip(seq[:-1], seq[1:], names):
    print("%-28s %7.0f ns" % (nm, ((p[:, b] - p[:, a]).mean()) * 10))
print("%-28s %7.0f ns" % ("S8 -> next frame top", ((p[1:, 0] - p[:-1, 10]).mean()) * 10))
print("frame %7.0f ns" % (((p[1:, 0] - p[:-1, 0]).mean()) * 10))
