import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ["FLOWTRON_MFMA"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import flowtron
from flowtron_amd import ops, pipeline
from oracle import synth
cfg = dict(synth.DEFAULT_MODEL_CONFIG, n_text=60)
m = flowtron.Flowtron(**cfg); m.load_state_dict(synth.make_state_dict(cfg, 4)); m = m.cuda().eval()
b = synth.make_batch(cfg, [70, 61, 33, 70, 9], [14, 12, 12, 7, 3], seed=4, with_prior=True)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
def d(a, c): return (a - c).abs().max().item() if a is not None else None
with torch.enable_grad():
    enc, in32 = m._encode(b["speaker_ids"], b["text"], b["in_lens"])
    out32 = ops.lens32(b["out_lens"])
    x = b["mel"].permute(2, 0, 1).contiguous()
    step = m.flows[0]
    os.environ["FLOWTRON_PIPELINE"] = "0"
    s1 = step(x, enc, in32, out32, b["attn_prior"]); s2 = step(x, enc, in32, out32, b["attn_prior"])
    print("seq vs seq   :", [d(a, c) for a, c in zip(s1, s2)])
    for graph in ("0", "1"):
        for ch in ("16", "35", "70"):
            os.environ.update(FLOWTRON_CHUNK=ch, FLOWTRON_LSTM_GRAPH=graph)
            p1 = pipeline.ar_step_forward_pipelined(step, x, enc, in32, out32, b["attn_prior"])
            p2 = pipeline.ar_step_forward_pipelined(step, x, enc, in32, out32, b["attn_prior"])
            torch.cuda.synchronize()
            print("graph", graph, "chunk", ch, "pipe vs pipe:", [d(a, c) for a, c in zip(p1, p2)], " pipe vs seq:", [d(a, c) for a, c in zip(p1, s1)])
    st = pipeline.lstm_state(step, "att", 70, 5, 1024, x.device)
    # stage outputs of the last pipelined run vs sequential recomputation of the attention LSTM
    a = step.attention_lstm
    mel0 = torch.cat([x.new_zeros(1, 5, 80), x[:-1]], 0)
    h_att = ops.lstm_layer(mel0, out32, a.weight_ih_l0, a.weight_hh_l0, a.bias_ih_l0, a.bias_hh_l0)
    print("h_att pipe-state vs seq:", d(st.y, h_att), "per-chunk:", [d(st.y[i:i+10], h_att[i:i+10]) for i in range(0, 70, 10)])
