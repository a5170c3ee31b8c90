"""Round-5 golden vectors from the REAL reference (/root/reference, CPU, build container only; never on the GPU box).

infer_depth.pt  Flowtron.infer (flowtron.py:901-930, :775-828) where round 4 stopped short: the decoder LSTM at n_lstm_layers = 1 and 3
                (flowtron.py:654-655 passes the depth to nn.LSTM; batch 1, gate disabled and enabled), and a BATCH of two equal-length
                utterances (no gate layer: `if sigmoid(gate) > thr` at :823 only works for one utterance) -- small model, 2 flows.
                Stored: the inferred mel, the attention rows of both flows, the number of frames with the gate on.
                Inputs and weights are rebuilt from seeds by oracle/synth.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import refshim, synth  # noqa: E402

CASES = [dict(name="depth1", n_lstm_layers=1, use_gate_layer=True, seed=41, n=24, in_len=9, batch=1),
         dict(name="depth3", n_lstm_layers=3, use_gate_layer=True, seed=43, n=24, in_len=7, batch=1),
         dict(name="batch2", n_lstm_layers=2, use_gate_layer=False, seed=45, n=20, in_len=8, batch=2)]


def case_inputs(case):
    cfg = dict(synth.SMALL_MODEL_CONFIG, n_lstm_layers=case["n_lstm_layers"], use_gate_layer=case["use_gate_layer"])
    sd = synth.make_state_dict(cfg, seed=case["seed"])
    rs = np.random.RandomState(case["seed"])
    B = case["batch"]
    residual = torch.from_numpy(rs.standard_normal((B, cfg["n_mel_channels"], case["n"])).astype(np.float32)) * 0.5
    text = torch.from_numpy(rs.randint(0, cfg["n_text"], (B, case["in_len"])))
    spk = torch.from_numpy(rs.randint(0, cfg["n_speakers"], (B,)))
    return cfg, sd, residual, spk, text


def run_case(R, case):
    cfg, sd, residual, spk, text = case_inputs(case)
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        mel, att = m.infer(residual.clone(), spk, text, gate_threshold=1.0)
        out = dict(case=case, mel=mel, attn=[torch.cat(a, 1) for a in att])       # per flow (reversed order): [B, N, L]
        if case["use_gate_layer"]:
            mel_g, _ = m.infer(residual.clone(), spk, text, gate_threshold=0.5)
            out["gated_frames"] = mel_g.shape[2]
    return out


def main():
    assert refshim.available(), "needs /root/reference"
    R = refshim.load()
    torch.manual_seed(0)
    res = [run_case(R, c) for c in CASES]
    for r in res:
        print(r["case"]["name"], tuple(r["mel"].shape), [tuple(a.shape) for a in r["attn"]], r.get("gated_frames"))
    torch.save({"cases": res}, os.path.join(HERE, "infer_depth.pt"))
    print("wrote infer_depth.pt")


if __name__ == "__main__":
    main()
