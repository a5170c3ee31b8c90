"""Round-4 golden: a multi-step TRAINING TRAJECTORY of the real reference (VERDICT r3 missing #2).

Run once in the build container:   python tests/golden/make_golden_r4.py

train_traj.pt   the REAL /root/reference modules (flowtron.Flowtron, flowtron.FlowtronLoss, radam.RAdam) driven on CPU through
                the exact statement sequence of train.py:282-331 for 5 iterations on two alternating batches:
                    model.zero_grad() -> forward -> criterion -> loss = nll + gate (+ ctc * w) -> loss.backward()
                    -> torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_val) -> optimizer.step()
                (fp16_run = False: the GradScaler calls of train.py:323-331 are pass-throughs), dropout neutralised
                (F.dropout -> identity, as in make_golden.py: the only stochastic op of the path).
                Stored: the four losses and the pre-clip gradient norm of every iteration, every parameter after the last
                iteration, and the optimizer's step count.  Inputs and initial weights are rebuilt from seeds by oracle/synth.py.
lstm_depth.pt   the decoder LSTM at the depths the config schema allows besides config.json's 2 (flowtron.py:655 passes n_lstm_layers to
                nn.LSTM): the real reference at n_lstm_layers = 1 and 3 (small model, ragged batch): z, the three losses, every gradient.
"""
import importlib.util
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import refshim, synth  # noqa: E402

TRAJ = dict(cfg=dict(synth.SMALL_MODEL_CONFIG), seed=21, iters=5, lr=3e-3, weight_decay=1e-6, grad_clip_val=1.0, ctc_loss_weight=0.01,
            blank_logprob=-8, sigma=1.0,
            batches=[dict(out_lens=[23, 19, 12, 23], in_lens=[9, 8, 8, 4], seed=21),
                     dict(out_lens=[17, 25, 25, 6], in_lens=[10, 7, 5, 3], seed=22)])


def run_reference_loop(Flowtron, FlowtronLoss, RAdam, device="cpu", neutralise_dropout=True, spec=TRAJ, prepare=None):
    """train.py:205-331 with n_gpus = 1, fp16_run = False; `Flowtron` / `FlowtronLoss` / `RAdam` are the classes to drive (the real
    reference here; the drop-in modules in tests/test_gpu_train_loop.py, which imports THIS function so both sides run the same
    statements).  `prepare(model, batch)`, if given, runs before every forward (the drop-in encoder draws its dropout keep-masks
    itself instead of calling F.dropout: the test hands it all-ones masks there)."""
    cfg = spec["cfg"]
    criterion = FlowtronLoss(spec["sigma"], bool(cfg["n_components"]), True, True, spec["ctc_loss_weight"], spec["blank_logprob"])
    model = Flowtron(**cfg)
    model.load_state_dict(synth.make_state_dict(cfg, seed=spec["seed"]))
    model = model.to(device)
    optimizer = RAdam(model.parameters(), lr=spec["lr"], weight_decay=spec["weight_decay"])
    batches = [synth.make_batch(cfg, b["out_lens"], b["in_lens"], seed=b["seed"], with_prior=True) for b in spec["batches"]]
    real_dropout = F.dropout
    if neutralise_dropout:
        F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    losses, norms = [], []
    try:
        model.train()
        for iteration in range(spec["iters"]):
            b = batches[iteration % len(batches)]
            model.zero_grad()
            mel, spk_ids, txt = b["mel"].clone().to(device), b["speaker_ids"].to(device), b["text"].to(device)
            in_lens, out_lens = b["in_lens"].to(device), b["out_lens"].to(device)
            gate_target = b["gate_target"].to(device)
            attn_prior = b["attn_prior"].clone().to(device)
            if prepare is not None:
                prepare(model, b)
            out = model(mel, spk_ids, txt, in_lens, out_lens, attn_prior)
            loss_nll, loss_gate, loss_ctc = criterion(out, gate_target, in_lens, out_lens, is_validation=False)
            loss = loss_nll + loss_gate
            loss += loss_ctc * criterion.ctc_loss_weight
            losses.append([loss.item(), loss_gate.item(), loss_nll.item(), loss_ctc.item()])
            loss.backward()
            total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), spec["grad_clip_val"])
            norms.append(float(total_norm))
            optimizer.step()
    finally:
        F.dropout = real_dropout
    return dict(losses=torch.tensor(losses, dtype=torch.float64), grad_norms=torch.tensor(norms, dtype=torch.float64),
                params={k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}, model=model, optimizer=optimizer)


DEPTH_CASES = [dict(n_lstm_layers=1, seed=31, out_lens=[19, 14, 19], in_lens=[8, 6, 5]),
               dict(n_lstm_layers=3, seed=33, out_lens=[21, 9, 16, 21], in_lens=[9, 7, 7, 3])]


def run_depth_case(R, case):
    cfg = dict(synth.SMALL_MODEL_CONFIG, n_lstm_layers=case["n_lstm_layers"])
    sd = synth.make_state_dict(cfg, seed=case["seed"])
    b = synth.make_batch(cfg, case["out_lens"], case["in_lens"], seed=case["seed"], with_prior=True)
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)                              # strict: the spec of oracle/synth.py matches the reference's registration
    crit = R.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    try:
        m.train()
        out = m(b["mel"].clone(), b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"].clone())
        z = out[0].detach().clone()
        nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        (nll + gl + 0.01 * ctc).backward()
    finally:
        F.dropout = real_dropout
    return dict(case=case, cfg=cfg, z=z, nll=nll.detach(), gate_loss=gl.detach(), ctc=ctc.detach(),
                grads={k: p.grad.detach().clone() for k, p in m.named_parameters()})


def main():
    assert refshim.available(), "needs /root/reference"
    R = refshim.load()
    torch.save({"cases": [run_depth_case(R, c) for c in DEPTH_CASES]}, os.path.join(HERE, "lstm_depth.pt"))
    print("wrote lstm_depth.pt")
    spec = importlib.util.spec_from_file_location("_ref_radam", os.path.join(refshim.REF_DIR, "radam.py"))
    radam = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(radam)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    res = run_reference_loop(R.Flowtron, R.FlowtronLoss, radam.RAdam)
    init = synth.make_state_dict(TRAJ["cfg"], seed=TRAJ["seed"])
    moved = {k: float((res["params"][k] - init[k]).abs().max()) for k in init}
    print("losses per iteration (total, gate, nll, ctc):\n", res["losses"])
    print("grad norms:", res["grad_norms"].tolist())
    print("largest weight movement: %.3e (%s)" % (max(moved.values()), max(moved, key=moved.get)))
    steps = sorted({int(st["step"]) for st in res["optimizer"].state.values() if "step" in st})
    torch.save({"spec": TRAJ, "losses": res["losses"], "grad_norms": res["grad_norms"], "params": res["params"], "optimizer_steps": steps},
               os.path.join(HERE, "train_traj.pt"))
    print("wrote train_traj.pt (%d parameters, optimizer steps %s)" % (len(res["params"]), steps))


if __name__ == "__main__":
    main()
