"""Round-2 golden vectors, again produced by running the REAL reference (/root/reference) on CPU.

Run once in the build container:   python tests/golden/make_golden_r2.py

Fixtures:
  radam_traj.pt  /root/reference/radam.py (RAdam) stepped 12 times on four small tensors with seeded gradients,
                 with and without torch's clip_grad_norm_(1.0) before the step (train.py:323-331), two weight decays:
                 parameters after every step + final exp_avg / exp_avg_sq.  Pins ft_sumsq + ft_radam_step, the N_sma >= 5
                 switch (step 6) and the checkpoint round trip.
  cfg_libritts.pt  BASELINE configs[2]-shaped case (tests/libri_case.py: 123 speakers, L = 237, ragged B = 4, H = 1024): the
                 reference's fp32 losses and sampled gradients.
  cfg2_bf16.pt   BASELINE config 2 model (2-flow LJS config.json defaults, H = 1024) on a B = 4, T <= 120 batch:
                 the reference in fp32 AND under torch.autocast("cpu", bfloat16) (the dtype the benchmark is quoted in;
                 train.py:292 wraps the forward in autocast).  Per parameter: fp32 gradient norm, a seeded sample of the
                 fp32 gradient, and the relative L2 deviation of the reference's OWN bf16-autocast gradient from its fp32
                 gradient -- the yardstick for the bf16-operand HIP path (are the 0.2-0.4 encoder-conv
                 gradient deviations a kernel property or conditioning?).
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import refshim, synth  # noqa: E402

RADAM_SHAPES = [(7, 5), (33,), (1,), (4, 3, 2)]
CFG2_LENS = ([120, 96, 80, 64], [24, 18, 15, 12])
CFG2_SEED = 77
N_SAMPLE = 8192


def radam_grads(step, shapes, seed=4242):
    """gradients of step `step` (1-based): shared by the generator and the tests."""
    g = torch.Generator().manual_seed(seed + step)
    scale = 0.05 if step % 3 else 3.0            # every third step is large enough for the clip to bite
    return [torch.randn(s, generator=g) * scale for s in shapes]


def radam_params(shapes, seed=4242):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in shapes]


def run_radam(clip, wd, n_steps=12):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_radam", os.path.join(refshim.REF_DIR, "radam.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ps = [torch.nn.Parameter(p.clone()) for p in radam_params(RADAM_SHAPES)]
    opt = mod.RAdam(ps, lr=1e-3, weight_decay=wd)
    traj = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(1, n_steps + 1):
            for p, g in zip(ps, radam_grads(step, RADAM_SHAPES)):
                p.grad = g.clone()
            if clip:
                torch.nn.utils.clip_grad_norm_(ps, clip)
            opt.step()
            traj.append([p.detach().clone() for p in ps])
    return {"clip": clip, "wd": wd, "traj": traj, "exp_avg": [opt.state[p]["exp_avg"].clone() for p in ps],
            "exp_avg_sq": [opt.state[p]["exp_avg_sq"].clone() for p in ps]}


def stable_key_seed(key):
    """hash() is salted per process: use a stable digest for the sampling seed."""
    import zlib
    return zlib.crc32(key.encode()) & 0x7FFFFFFF


def sample_idx_stable(numel, key):
    if numel <= N_SAMPLE:
        return None
    rs = np.random.RandomState(stable_key_seed(key))
    return torch.from_numpy(np.sort(rs.choice(numel, N_SAMPLE, replace=False)))


def run_cfg2(R, autocast):
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    sd = synth.make_state_dict(cfg, seed=CFG2_SEED)
    b = synth.make_batch(cfg, CFG2_LENS[0], CFG2_LENS[1], seed=CFG2_SEED, with_prior=True)
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)
    m.train()
    crit = R.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    real = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = m(b["mel"].clone(), b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"].clone())
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        (nll + gl + 0.01 * ctc).backward()
    finally:
        F.dropout = real
    return (nll.detach().float(), gl.detach().float(), ctc.detach().float()), {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}


def run_libritts(R):
    """cfg_libritts.pt: the real reference (fp32) on tests/libri_case.py: losses + per parameter the gradient norm and a seeded
    sample of the gradient (same sampling as cfg2_bf16.pt)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import libri_case
    cfg, sd, b = libri_case.make()
    m = R.Flowtron(**cfg)
    m.load_state_dict(sd)
    m.train()
    crit = R.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    real = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x.clone()
    try:
        out = m(b["mel"].clone(), b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"].clone())
        nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
        (nll + gl + 0.01 * ctc).backward()
    finally:
        F.dropout = real
    res = {"losses_fp32": (nll.detach().float(), gl.detach().float(), ctc.detach().float()), "grad": {}}
    for k, p in m.named_parameters():
        flat = p.grad.detach().float().reshape(-1)
        idx = sample_idx_stable(flat.numel(), k)
        res["grad"][k] = {"norm": flat.norm().item(), "idx": idx, "sample": (flat if idx is None else flat[idx]).clone()}
    torch.save(res, os.path.join(HERE, "cfg_libritts.pt"))
    print("cfg_libritts.pt", os.path.getsize(os.path.join(HERE, "cfg_libritts.pt")) // 1024, "KiB; losses", [x.item() for x in res["losses_fp32"]])


def main():
    assert refshim.available(), "needs /root/reference"
    torch.set_num_threads(8)
    if "--libritts-only" in sys.argv:
        run_libritts(refshim.load())
        return
    torch.save({"shapes": RADAM_SHAPES, "cases": [run_radam(c, wd) for c in (0.0, 1.0) for wd in (1e-6, 1e-2)]},
               os.path.join(HERE, "radam_traj.pt"))
    R = refshim.load()
    l32, g32 = run_cfg2(R, False)
    l16, g16 = run_cfg2(R, True)
    res = {"seed": CFG2_SEED, "out_lens": CFG2_LENS[0], "in_lens": CFG2_LENS[1], "losses_fp32": l32, "losses_bf16_autocast": l16,
           "grad": {}}
    for k in g32:
        idx = sample_idx_stable(g32[k].numel(), k)
        flat = g32[k].reshape(-1)
        nrm = flat.norm().item()
        dev = (g16[k].reshape(-1) - flat).norm().item() / max(nrm, 1e-30)
        res["grad"][k] = {"norm": nrm, "ref_bf16_autocast_rel_dev": dev, "idx": idx,
                          "sample": (flat if idx is None else flat[idx]).clone()}
        print("%-58s |g| %.3e  reference bf16-autocast rel dev %.4f" % (k, nrm, dev))
    torch.save(res, os.path.join(HERE, "cfg2_bf16.pt"))
    for f in ("radam_traj.pt", "cfg2_bf16.pt"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
    run_libritts(R)


if __name__ == "__main__":
    main()
