#!/usr/bin/env python3
"""Benchmark of the Flowtron hot path on MI355X (contract: see the task description / DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = zero_grad + Flowtron.forward + FlowtronLoss (NLL + gate + attention-CTC) + backward (with ONE in-place RCCL
all-reduce(AVG) of the flat gradient arena at the end of backward when N > 1; per-flow buckets under backward with FLOWTRON_DP_OVERLAP=1) + global-norm clip + fused RAdam update, on BASELINE.json configs[1]:
2-flow LJS config.json model, 80-bin mels, per-GPU batch 32 of LJSpeech-shaped synthetic utterances (<= 10 s),
attention prior + CTC on, bf16 MFMA operands with fp32 accumulate/storage.  Weak scaling: every rank processes its
own 32 utterances.  value = valid mel frames (sum of out_lens over all ranks and steps) / wall time.

Rank 0 prints ONE JSON line.  `roofline` = the whole step against the MFMA roof; under it `dominant_kernel` / `second_kernel`
(the persistent LSTM recurrences, timed live with HIP events on the launch stream next to the launch-per-step kernels they
replace) carry their own bound: a per-step hand-off-latency floor, with the HBM and MFMA fractions beside it; `parity` (the HIP gradients of a batch slice against the CPU oracle), `cpu_baseline` (the oracle timed on this box's host
cores on a bounded sample), `infer` (RTF of a 400-frame 2-flow decode with 16-bit weight images, median of 7 calls) and
`infer_fp32` (the same decode with fp32 weights and arithmetic -- the precision of the reference's inference.py).

--config libritts / libritts_fp16 run BASELINE configs[2] / configs[4] (123 speakers, texts up to 237 symbols; fp16 MFMA
operands + torch GradScaler, no attention prior) instead; --mfma overrides the operand type (bf16 | f16 | f32).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_CONFIG = {   # reference config.json:49-66 (LJS defaults)
    "n_speakers": 1, "n_speaker_dim": 128, "n_text": 185, "n_text_dim": 512, "n_flows": 2, "n_mel_channels": 80,
    "n_attn_channels": 640, "n_hidden": 1024, "n_lstm_layers": 2, "mel_encoder_n_hidden": 512, "n_components": 0,
    "mean_scale": 0.0, "fixed_gaussian": True, "dummy_speaker_embedding": False, "use_gate_layer": True,
    "use_cumm_attention": False,
}
HOP, SR = 256, 22050


def synth_batch(B, seed, t_max=862, l_max=187, n_text=185, l_min=12, n_speakers=1, chars_per_frame=1 / 5.5):
    """LJSpeech-shaped synthetic batch: out_lens ~ clip(N(566,190),100,862), text length ~
    frames/5.5, sorted by text length (data.py:200-202), log-mel-range values, beta-binomial attention prior.
    SURVEY 8d configs 3 / 5 (LibriTTS): n_speakers = 123 with uniform speaker ids, text lengths clipped to [5, 237]
    (chars_per_frame raised so the batch really reaches L = 237: two 128-column score tiles, 475 CTC states)."""
    rs = np.random.RandomState(seed)
    out = np.clip(np.round(rs.normal(566, 190, B)), 100, t_max).astype(int)
    out[0] = t_max
    inn = np.clip(np.round(out * chars_per_frame), l_min, l_max).astype(int)
    order = np.argsort(-inn, kind="stable")
    out, inn = out[order], inn[order]
    T, L = int(out.max()), int(inn.max())
    mel = np.zeros((B, 80, T), np.float32)
    text = np.zeros((B, L), np.int64)
    gate = np.zeros((B, T), np.float32)
    for b in range(B):
        t, l = int(out[b]), int(inn[b])
        base = -5.0 + 2.0 * np.sin(np.linspace(0, 6.0, 80))[:, None]
        walk = np.cumsum(0.15 * rs.standard_normal((1, t)), axis=1)
        mel[b, :, :t] = np.clip(base + walk + 1.2 * rs.standard_normal((80, t)), -11.5, 1.0)
        text[b, :l] = rs.randint(0, n_text, size=l)
        gate[b, t - 1:] = 1.0
    return dict(mel=torch.from_numpy(mel), text=torch.from_numpy(text), gate=torch.from_numpy(gate),
                speaker_ids=torch.from_numpy(rs.randint(0, n_speakers, size=B).astype(np.int64)) if n_speakers > 1
                else torch.zeros(B, dtype=torch.long), in_lens=torch.from_numpy(inn.astype(np.int64)),
                out_lens=torch.from_numpy(out.astype(np.int64)))


def beta_binomial_prior_batch(in_lens, out_lens, T, L):
    """data.py:31-41 closed form (lgamma), float64 on CPU once -- data-pipeline work, outside the timed region."""
    pr = torch.zeros(len(in_lens), T, L)
    lg = torch.lgamma
    for b, (P, M) in enumerate(zip(in_lens.tolist(), out_lens.tolist())):
        k = torch.arange(P, dtype=torch.float64)[None, :]
        i = torch.arange(1, M + 1, dtype=torch.float64)[:, None]
        a, bb = i, (M + 1 - i)
        n = float(P - 1)
        logp = (lg(torch.tensor(n + 1, dtype=torch.float64)) - lg(k + 1) - lg(n - k + 1) + lg(k + a) + lg(n - k + bb)
                - lg(n + a + bb) - (lg(a) + lg(bb) - lg(a + bb)))
        pr[b, :M, :P] = torch.exp(logp).float()
    return pr


def init_weights(model, seed):
    """random-init weights of the named architecture; the coupling conv is zero-initialised by the reference
    (flowtron.py:651-653), which would make log_s == 0 -- give it small random values so the NLL is non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("conv.weight") and "convolutions" not in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.01)
            elif name.endswith("conv.bias") and "convolutions" not in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def cumm_roofline(B, in_lens_cpu, mode, frames=64):
    """--config ljs_cumm: the fused frame kernels of cumulative attention (csrc/cumm_fused.hip: ONE launch per frame and direction),
    timed live with HIP events on the launch stream around a forward and a backward call of ops.CummAttnSeqFn at the bench's own
    B / L / text lengths over `frames` frames (a frame's cost does not depend on T).  Per frame and per workgroup the kernels pull
    their weight fragments from the L2 (no reuse across frames: a launch keeps nothing), so the roof that binds them is the L2 ->
    CU path: 34.5 TB/s over 256 CUs = 135 GB/s per CU (MI355X_MICROARCH.md); MFMA and HBM fractions stand beside it.
    Bytes per workgroup and frame (16-bit fragments): forward = W_key rows of its column half (A E 2 / 2) + w2 (E 96 2) + its lane-order
    text tile (32 E 4); backward = W_key^T (E A 2) + w2 (E 96 2) + w2^T + text and dtext tiles (3 x 32 E 4, the dtext tile written back)."""
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    E = A = 640
    Lk = int(in_lens_cpu.max())
    T = frames
    f = dict(device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *sh: torch.randn(*sh, generator=g, **f)
    lens = in_lens_cpu.to(device="cuda", dtype=torch.int32)
    leaves = [t.requires_grad_(True) for t in (rn(T, B, A) * 0.7, rn(Lk, B, A), rn(Lk, B, E) * 0.7, rn(A, E) / E ** 0.5, rn(1, A) / A ** 0.5 * 4,
                                               rn(32, 2, 5) * 0.5, rn(32) * 0.2, rn(E, 32, 3) * 0.2, rn(E) * 0.2)]
    ts = []
    for rep in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        c, a_, lp = ops.CummAttnSeqFn.apply(*leaves, lens, 1.0, mode)
        e1.record()
        (c.sum() + (a_ * a_).sum()).backward()
        e2.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) * 1e3 / T, e1.elapsed_time(e2) * 1e3 / T))
    fwd_us, bwd_us = sorted(t[0] for t in ts)[1], sorted(t[1] for t in ts)[1]
    rows = int(in_lens_cpu.sum())
    tiles_f = int(sum(-(-int(l) // 32) for l in in_lens_cpu))
    tiles_b = int(sum(int(l) // 26 + 1 for l in in_lens_cpu))
    split = 2 * tiles_f <= 256
    wg_f, wg_b = (2 if split else 1) * tiles_f, tiles_b
    by_f = A * E * 2 // (2 if split else 1) + E * 96 * 2 + 32 * E * 4
    by_b = E * A * 2 + 2 * E * 96 * 2 + 3 * 32 * E * 4
    fl_f = 2.0 * rows * (96 * E + E * A)                       # cond + key projection, valid rows
    fl_b = 2.0 * rows * (96 * E + A * E + E * 96) + 2.0 * rows * (A * E + E * 97)   # in-frame GEMMs + this frame's share of the chunk GEMMs
    hbm_f = rows * A * 4
    hbm_b = rows * A * 4 + rows * (A + 2 * E + 128) * 2
    def pmc(name, what):
        """committed PMC pass over scripts/exp/cumm_prof.py (same B / L, lengths 60..157): per-launch average of a counter"""
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_cumm_%s.json" % what)))
            for k, v in d.items():
                if name in k:
                    key = {"FETCH_SIZE": "FETCH_SIZE", "WRITE_SIZE": "WRITE_SIZE", "MFMA_BUSY": "mfma_busy_frac"}[what]
                    x = v[key]
                    return float(x["avg"] if isinstance(x, dict) else x)
        except Exception:
            pass
        return None
    out = {}
    for name, us, wgs, by, fl, hbm in (("cummf_fwd_k", fwd_us, wg_f, by_f, fl_f, hbm_f), ("cummf_bwd_k", bwd_us, wg_b, by_b, fl_b, hbm_b)):
        per_cu = by / (us * 1e-6) / 1e9
        fs, ws, mb = pmc(name, "FETCH_SIZE"), pmc(name, "WRITE_SIZE"), pmc(name, "MFMA_BUSY")
        out[name] = {"kernel": name + "<10, 10>", "bound": "l2", "achieved": round(per_cu, 1), "peak": 135.0, "unit": "GB/s per CU",
                     "frac": round(per_cu / 135.0, 3), "us_per_frame_incl_launch": round(us, 2), "workgroups_per_frame": wgs,
                     "l2_bytes_per_workgroup_per_frame": by,
                     "l2_chip": {"achieved": round(by * wgs / (us * 1e-6) / 1e12, 2), "peak": 34.5, "unit": "TB/s"},
                     "mfma": {"achieved": round(fl / (us * 1e-6) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / (us * 1e-6) / 2.5e15, 4),
                              "flop_per_frame": fl},
                     "hbm": {"achieved": round(hbm / (us * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm / (us * 1e-6) / 8e12, 4),
                             "bytes_per_frame": hbm,
                             "traffic": None if fs is None or ws is None else int((2 * fs + ws) * 1024), "pmc_round": "r05 (profiles/r05_pmc_cumm_*.json)"},
                     "mfma_busy_frac_pmc": None if mb is None else round(mb, 4),
                     "note": "one launch per frame; us_per_frame includes the launch boundary (events around the whole call / frames; the "
                             "backward figure also carries the chunk's weight-gradient GEMMs); stage stamps of a workgroup: "
                             "profiles/r05_*_cumm_stage_stamps.log"}
    return out


def lstm_step_roofline(B, H, T, mode):
    """Live timing of the dominant kernel (lstm_fwd_step, flowtron_amd/csrc/lstm.hip): T launches bracketed by HIP
    events on the launch stream.  Algorithmic bytes per launch = W_hh (4H*H) + h_prev (B*H) + gate pre-activations in
    (B*4H) + y, h, c, cell out (4*B*H) + saved gates (B*4H), all fp32 (kernel table)."""
    from flowtron_amd import _lib as L
    dev = "cuda"
    gx = torch.randn(T, B, 4 * H, device=dev) * 0.1
    w = torch.randn(4 * H, H, device=dev) / H ** 0.5
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    y = torch.empty(T, B, H, device=dev)
    gates = torch.empty(T, B, 4 * H, device=dev)
    cell = torch.empty(T, B, H, device=dev)
    work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)

    def run():
        L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(work),
                                        T, B, H, 0, mode, L.stream()), "ft_lstm_seq_fwd")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / T
    wb = 2 if mode == 1 else 4      # bf16 path streams W_hh and h_{t-1} as bf16 fragment images
    # per launch: W_hh + h_{t-1} in; gx row in (fp32); y, saved gates, saved cell out (fp32); cell state in+out (fp32);
    # h_t out in the operand format
    bytes_per_launch = wb * (4 * H * H + B * H) + 4 * (B * 4 * H) + 4 * (B * H + B * 4 * H + B * H) + 4 * (2 * B * H) + wb * B * H
    achieved = bytes_per_launch / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "lstm_fwd_step", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 4), "traffic": pmc_traffic("lstm_fwd_step"), "us_per_launch": round(us, 3),
            "bytes_per_launch": bytes_per_launch}


def lstm2_step_roofline(B, H, T):
    """Live timing of THE dominant kernel of the step (lstm2_fwd_both, flowtron_amd/csrc/lstm2.hip: ~22 % of the step):
    T+1 launches of the two-layer wavefront chain bracketed by HIP events on the launch stream.  Algorithmic bytes per
    launch (steady state, DESIGN.md kernel table): bf16 fragment images of W_hh0 and [W_ih1|W_hh1] (4H*H*2 + 4H*2H*2),
    bf16 images of h0[s-1] and h1[s-2] in, gx0 row + bias1 in (fp32), cell state of both layers in+out (fp32), and per layer
    y, saved gates, saved cell (fp32) + the bf16 image of h out."""
    from flowtron_amd import _lib as L
    dev = "cuda"
    f = dict(device=dev, dtype=torch.float32)
    gx = torch.randn(T, B, 4 * H, **f) * 0.1
    w0, wi1, w1 = (torch.randn(4 * H, H, **f) / H ** 0.5 for _ in range(3))
    b1 = torch.zeros(4 * H, **f)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    y0, y1 = torch.empty(T, B, H, **f), torch.empty(T, B, H, **f)
    g0, g1 = torch.empty(T, B, 4 * H, **f), torch.empty(T, B, 4 * H, **f)
    c0, c1 = torch.empty(T, B, H, **f), torch.empty(T, B, H, **f)
    work = torch.empty(L.lib().ft_lstm2_workspace_bytes(B, H), device=dev, dtype=torch.uint8)

    def run():
        L.check(L.lib().ft_lstm2_seq_fwd(L.ptr(gx), L.ptr(w0), L.ptr(wi1), L.ptr(b1), L.ptr(w1), L.ptr(lens), L.ptr(y0), L.ptr(g0),
                                         L.ptr(c0), L.ptr(y1), L.ptr(g1), L.ptr(c1), L.ptr(work), T, B, H, L.stream()), "ft_lstm2_seq_fwd")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (T + 1)
    bp = (B + 15) // 16 * 16
    bytes_per_launch = (2 * (4 * H * H + 4 * H * 2 * H)            # weight images
                        + 2 * 2 * bp * H                           # h0[s-1], h1[s-2] images in
                        + 4 * (B * 4 * H + 4 * H)                  # gx0 row, bias1
                        + 2 * 2 * 4 * B * H                        # cell state of both layers in + out
                        + 2 * (4 * (B * H + B * 4 * H + B * H) + 2 * bp * H))   # per layer: y, gates, cell out + h image out
    achieved = bytes_per_launch / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "lstm2_fwd_both", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 4), "traffic": pmc_traffic("lstm2_fwd_both"), "us_per_launch": round(us, 3),
            "bytes_per_launch": bytes_per_launch}


def persist_roofline(B, H, T, lens_cpu, mode=1):
    """Live timing of the dominant kernels of the step when the persistent recurrence path is active, launched the way the step
    launches them (ops.LSTMSeqFn / ops.DecoderPairFn), HIP events on the launch stream around one sequence of the bench's own
    shape and lengths:
      backward  lstm_persist_bwd_rs_k (csrc/lstm_persist.hip): one launch per LSTM and sequence, 6 per step, through the entry
                point the step uses (ft_lstm_persist_bwd_img, image only);
      forward   lstm_roles_fwd_k (csrc/lstm_roles.hip, round 6): the attention LSTM as ONE launch at 4 rows per XCD group, and the
                decoder layer pair as a pipeline of n + 1 role launches over n time chunks (first / last at 4 rows, the n - 1
                between them two roles at 8 rows per group) -- 2 x (1 + n + 1) launches per step.
    Algorithmic HBM bytes per launch: the 16-bit fragment image of W_hh once (4H*H*2) + per VALID (t, b) row the fp32
    rows the recurrence must read and write -- forward: gx row in (4H, 16-bit since round 6) + y, saved gates, saved cell out
    (H + 4H + H, fp32); backward: saved gates, cell, dy in (4H + H + H) + dgates out (4H, 16-bit in the image).  The kernels are bound by the per-step dependency
    (an L2 hand-off + the cell update per step), not by bandwidth: `frac` is small by construction and `us_per_step` is the figure
    of merit; the launch-per-step kernels they replace are timed beside them."""
    from flowtron_amd import _lib as L
    from flowtron_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    f = dict(device=dev, dtype=torch.float32)
    torch.manual_seed(0)
    gx = torch.randn(T, B, 4 * H, **f) * 0.5
    w = torch.randn(4 * H, H, **f) / H ** 0.5
    dy = torch.randn(T, B, H, **f) * 0.1
    lens = lens_cpu.to(device=dev, dtype=torch.int32)
    y, gates, cell, dgx = torch.empty(T, B, H, **f), torch.empty(T, B, 4 * H, **f), torch.empty(T, B, H, **f), torch.empty(T, B, 4 * H, **f)
    y2, gates2, cell2 = torch.empty(T, B, H, **f), torch.empty(T, B, 4 * H, **f), torch.empty(T, B, H, **f)
    wp = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
    ws = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dev, dtype=torch.uint8)
    st = ops.persist_status(dev)
    ng_bwd = ops.PERSIST_BWD_CODE
    lib = L.lib()
    rm = ops.RowMap(lens, T, B)
    dimg = ops.Bf16Image.empty_rows(4 * H, rm, mode, dev)
    img_only = os.environ.get("FLOWTRON_LSTM_PERSIST_IMG", "1") == "1"
    wimg = ops.roles_wimg(w, mode, False)
    gx16 = ops._GX16 and L.is16(mode)              # the step's projections hand gx over as 16-bit rows (ops.gx16_ok)
    if gx16:
        gx = gx.to(ops.op16_dtype(mode))
    gx32 = gx.float()                              # (the launch-per-step yardstick reads fp32 rows)
    n_pair = ops.decoder_pair_chunks(B, H, mode, dev, T)
    edges = ops._chunk_edges(T, n_pair) if n_pair else None
    s1, s2 = torch.zeros(2, B, H, **f), torch.zeros(2, B, H, **f)

    def pair_pipeline():
        for k in range(n_pair + 1):
            roles = []
            if k < n_pair:
                roles.append(ops.fwd_role(gx, lens, y, gates, cell, wimg, edges[k], edges[k + 1], s1))
            if k > 0:
                roles.append(ops.fwd_role(gx, lens, y2, gates2, cell2, wimg, edges[k - 1], edges[k], s2))
            ops.roles_launch(roles, 8 if len(roles) == 2 else 4, mode, dev)

    runs = {
        "lstm_roles_fwd_k": lambda: ops.roles_launch([ops.fwd_role(gx, lens, y, gates, cell, wimg)], 4, mode, dev),
        "lstm_persist_bwd_k": (lambda: L.check(L.op16("ft_lstm_persist_bwd_img", mode)(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), None,
                                                                        L.ptr(wp), L.ptr(st), T, B, H, ng_bwd, L.ptr(dimg.buf), dimg.ld,
                                                                        dimg.buf.numel() // (2 * dimg.ld), L.ptr(dimg.colsum), L.stream()), "persist bwd img"))
        if img_only else
        (lambda: L.check(L.op16("ft_lstm_persist_bwd", mode)(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx),
                                                                        L.ptr(wp), L.ptr(st), T, B, H, ng_bwd, L.stream()), "persist bwd")),
        "lstm_fwd_step": lambda: L.check(lib.ft_lstm_seq_fwd(L.ptr(gx32), L.ptr(w), L.ptr(lens), L.ptr(y), H, L.ptr(gates), L.ptr(cell), L.ptr(ws),
                                                              T, B, H, 0, mode, L.stream()), "step fwd"),
        "lstm_bwd_step_bf16": lambda: L.check(lib.ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx),
                                                                   L.ptr(ws), T, B, H, 0, mode, L.stream()), "step bwd"),
    }
    if n_pair:
        runs["pair_pipeline"] = pair_pipeline
    us = {}
    for name, fn in runs.items():
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us[name] = sorted(ts)[1]
    ops.check_persist_status()
    rows = int(lens_cpu.sum())
    hops = handoff_hops()
    out = {}
    # What bounds a step of these kernels is the dependency chain, not HBM and not the MFMA rate.  floor_us_per_step = the part of
    # that chain no schedule can remove, from the kernel's structure and measured primitives:
    #   one same-XCD L2 hand-off per step (publish -> every consumer sees the data): profiles/r02_handoff_hops.json;
    #   the MFMAs one wave must issue back to back per step (64 x v_mfma_f32_16x16x32: 7.3 ns each from one wave per SIMD, measured:
    #   scripts/exp/mfma_rate_probe.hip);
    #   the hand-off bytes every CU pulls from its XCD's L2 per step (forward: 8 KB of bare operand pairs at 4 rows per group;
    #   backward: 16 + 16 KB of fp32 partials in reduce-scatter form; x 256 CUs) at the measured L2 peak of 34.5 TB/s.
    # The LDS reduce, the barrier, the cell update and the skew between the 32 CUs of a group are what `floor_frac` leaves.
    mfma_us = 0.47
    bwd_out = 2 * 4 * H if img_only else 4 * 4 * H
    for name, kname, per_row, repl, gran_kb in (("lstm_persist_bwd_k", "lstm_persist_bwd_rs_k", 4 * (4 * H + H + H) + bwd_out, "lstm_bwd_step_bf16", 32),
                                                ("lstm_roles_fwd_k", "lstm_roles_fwd_k<4, false, %s>" % ("true" if gx16 else "false"), (2 if gx16 else 4) * 4 * H + 4 * (H + 4 * H + H), "lstm_fwd_step", 8)):
        nbytes = 2 * 4 * H * H + rows * per_row
        ach = nbytes / (us[name] * 1e-6) / 1e9
        per_step = us[name] / T
        l2_us = 256 * gran_kb * 1024 / 34.5e12 * 1e6
        floor = hops["same_xcd_hop_us"] + mfma_us + l2_us
        mb, mb_src = pmc_value("MFMA_BUSY", kname, "mfma_busy_frac", with_source=True)
        # hardware roofs: algorithmic MFMA work = 2 * 4H * H flop per VALID (t, b) row (8.39 MFLOP at H 1024) against the
        # dense 16-bit peak, and the algorithmic HBM bytes above against 8 TB/s.  `frac` = the larger of the two -- the fraction of
        # the nearest HARDWARE roof; the latency-floor fraction of this design stands beside it as `floor_frac`.
        flops = 2.0 * 4 * H * H * rows
        tf = flops / (us[name] * 1e-6) / 1e12
        mfma_frac, hbm_frac = tf / 2500.0, ach / 8000.0
        rows_per_group = max(1, (B + 7) // 8)
        out[name] = {"kernel": kname, "bound": "hbm" if hbm_frac >= mfma_frac else "mfma",
                     "achieved": round(ach, 1) if hbm_frac >= mfma_frac else round(tf, 1),
                     "peak": 8000.0 if hbm_frac >= mfma_frac else 2500.0, "unit": "GB/s" if hbm_frac >= mfma_frac else "TFLOP/s",
                     "frac": round(max(hbm_frac, mfma_frac), 4),
                     "mfma": {"achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(mfma_frac, 4),
                              "flop_per_launch": flops, "note": "valid (t, b) rows x 2 x 4H x H"},
                     "hbm": {"achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm_frac, 4),
                             "bytes_per_launch": nbytes, "traffic": pmc_traffic(kname, nbytes)},
                     "mfma_rows_used": "%d/16" % rows_per_group,
                     "us_per_step": round(per_step, 3), "floor_us_per_step": round(floor, 3),
                     "floor_frac": round(floor / per_step, 3), "floor_bound": "handoff-latency",
                     "floor_terms_us": {"l2_handoff_hop": round(hops["same_xcd_hop_us"], 3), "mfma_issue_64_per_wave": round(mfma_us, 3),
                                        "granule_bytes_over_l2_peak": round(l2_us, 3)},
                     "steps_per_launch": T, "us_per_launch": round(us[name], 1),
                     "replaces": {"kernel": repl, "us_per_step": round(us[repl] / T, 3)},
                     "mfma_busy_frac_pmc": mb, "pmc_round": mb_src,
                     "note": "one launch = one whole sequence of T dependent steps, W_hh resident in registers, 8 batch groups (one per XCD) "
                             "of B / 8 rows each: only mfma_rows_used of an MFMA tile's 16 rows carry batch rows.  frac = achieved / peak "
                             "of the nearer HARDWARE roof (recompute: flop_per_launch or bytes_per_launch / us_per_launch); floor_frac = "
                             "this design's per-step hand-off-latency floor (floor_terms_us) / us_per_step -- a description of the design, "
                             "not of the hardware"}
    bwd, fwd = out["lstm_persist_bwd_k"], out["lstm_roles_fwd_k"]
    bwd["transport"] = ng_bwd
    bwd["entry_point"] = "ft_lstm_persist_bwd_img (image only)" if img_only else "ft_lstm_persist_bwd"
    bwd["launches_per_step"] = 6
    bwd["us_per_training_step"] = round(6 * us["lstm_persist_bwd_k"], 1)
    fwd["entry_point"] = "ft_lstm_roles_fwd (one role, 4 rows per XCD group): the attention LSTM's launch"
    if n_pair:
        # the decoder layer pair: two recurrences in n + 1 launches; the figure that compares with 2 x us_per_step of single launches
        fwd["decoder_pair_pipeline"] = {
            "chunks": n_pair, "launches": n_pair + 1, "us_per_pipeline": round(us["pair_pipeline"], 1),
            "us_per_pair_step": round(us["pair_pipeline"] / T, 3), "two_single_launches_us_per_pair_step": round(2 * us["lstm_roles_fwd_k"] / T, 3),
            "kernel": "lstm_roles_fwd_k<8, ..> (two roles: layer 0 on XCDs 0-3, layer 1 one chunk behind on XCDs 4-7; mfma_rows_used 8/16) "
                      "between one lstm_roles_fwd_k<4, ..> launch at either end",
            "note": "recurrence launches only; the chunks' input-projection GEMMs (ops.DecoderPairFn) run between them in the step"}
        fwd["launches_per_step"] = 2 * (1 + n_pair + 1)
        fwd["us_per_training_step"] = round(2 * (us["lstm_roles_fwd_k"] + us["pair_pipeline"]), 1)
    else:
        fwd["launches_per_step"] = 6
        fwd["us_per_training_step"] = round(6 * us["lstm_roles_fwd_k"], 1)
    return bwd, fwd


def handoff_hops():
    """measured hand-off hops (us) committed under profiles/ (scripts/exp/handoff_probe.hip); fixed fall-back = the same numbers"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_handoff_hops.json")))["used_by_bench"]
        return {"same_xcd_hop_us": float(d["same_xcd_hop_us"]), "cross_xcd_hop_us": float(d["cross_xcd_hop_us"])}
    except Exception:
        return {"same_xcd_hop_us": 0.2985, "cross_xcd_hop_us": 0.6432}


PMC_PREFIXES = ("r06_pmc_", "r05b_pmc_", "r05_pmc_", "r04_pmc_", "r03_pmc_", "r02_pmc_", "r01_pmc_lstm_")


def _pmc_file(prefix, counter_file):
    path = os.path.join(ROOT, "profiles", "%s%s.json" % (prefix, counter_file))
    return json.load(open(path)) if os.path.exists(path) else None


def _pmc_key_matches(key, kernel):
    """the summary's kernel name, stripped of `void ` and namespaces, IS `kernel`, or begins with it up to its template arguments /
    parameter list (`lstm_persist_bwd_rs_k` names every instantiation; `lstm_persist_bwd_rs_k<2, false>` exactly one): a bare
    substring test let `lstm_persist_fwd_k` pick up `bilstm_persist_fwd_k`, a bare prefix test one instantiation's counters for
    another's"""
    k = key.replace("void ", "").replace("(anonymous namespace)::", "")
    return k == kernel or (k.startswith(kernel) and k[len(kernel)] in "<(")


def _pmc_scalar(v, stat="avg"):
    """a field of scripts/pmc_summarize.py's output: either a number or {"avg": number, "dispatches": n, "max": largest dispatch}"""
    if isinstance(v, dict):
        return float(v[stat]) if stat in v else float(v["avg"])
    return float(v)


def pmc_value(counter_file, kernel_substr, field, with_source=False):
    """one derived field of the committed rocprofv3 --pmc summaries (profiles/rNN_pmc_<counter_file>.json), newest round first.
    A file that exists but cannot be read as expected RAISES (a swallowed TypeError made the line quote round-2
    counters for round-3 kernels); a kernel that a newer round's file does not list falls through to the older file, and the
    round the value comes from travels with it (`with_source`)."""
    for prefix in PMC_PREFIXES:
        d = _pmc_file(prefix, counter_file)
        if d is None:
            continue
        for k, v in d.items():
            if _pmc_key_matches(k, kernel_substr) and field in v:
                val = round(_pmc_scalar(v[field]), 4)
                return (val, prefix.split("_pmc")[0]) if with_source else val
    return (None, None) if with_source else None


def pmc_traffic(kernel_name, algorithmic_bytes=None):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (separate FETCH_SIZE and WRITE_SIZE
    runs, profiles/rNN_pmc_*.json, newest round first), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
    reads 1/2 of a wide coalesced 16 B/lane stream; counters are KiB).  bench.py cannot run rocprofv3 on itself, so this is
    the last measured value, or null when no round lists the kernel.  Both counters come from the SAME kernel entry (the first key
    that matches, of the instantiation with the most dispatches when a bare name matches several -- the step's kernel, not a
    self-test's; of its dispatches the largest one), and a figure below 0.9 x the algorithmic bytes is refused: the counters then
    belong to another launch shape."""
    for prefix in PMC_PREFIXES:
        fd, wd = _pmc_file(prefix, "FETCH_SIZE"), _pmc_file(prefix, "WRITE_SIZE")
        if fd is None or wd is None:
            continue
        keys = [k for k in fd if _pmc_key_matches(k, kernel_name) and "FETCH_SIZE" in fd[k] and k in wd and "WRITE_SIZE" in wd[k]]
        if not keys:
            continue

        def dispatches(k):
            v = fd[k]["FETCH_SIZE"]
            return v.get("dispatches", 1) if isinstance(v, dict) else fd[k].get("dispatches", 1)
        k = max(keys, key=dispatches)
        # the LARGEST dispatch of the kernel: the step also launches the roles kernels over time windows (the ends of the decoder pair
        # pipeline), and the figure is held against the bytes of a whole-sequence launch
        val = int((2.0 * _pmc_scalar(fd[k]["FETCH_SIZE"], "max") + _pmc_scalar(wd[k]["WRITE_SIZE"], "max")) * 1024)
        if algorithmic_bytes is not None and val < 0.9 * algorithmic_bytes:
            return None
        return val
    return None


def usable_cores():
    """Host cores this process may actually use: affinity mask, further limited by a cgroup CPU quota (a container
    that reports 200 cpus but is throttled to 16 would otherwise oversubscribe OpenMP by 10x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


ILL_CONDITIONED = ("encoder.convolutions", "embedding.weight", "attention_layer.query")   # tests/test_gpu_bench_path.py


CPU_SAMPLE_UTTS = 16          # utterances of the bounded CPU-oracle sample (cpu_baseline and parity legs)


def sample_of(batch, n_utt=CPU_SAMPLE_UTTS):
    """The bounded sample both the CPU oracle and the HIP parity pass evaluate: the n shortest utterances of the batch,
    re-sorted by text length (data.py:200-202), trimmed to their own T / L, with their beta-binomial prior."""
    idx = torch.argsort(batch["out_lens"])[:n_utt]
    idx = idx[torch.argsort(batch["in_lens"][idx], descending=True)]
    out_lens, in_lens = batch["out_lens"][idx], batch["in_lens"][idx]
    T, Lk = int(out_lens.max()), int(in_lens.max())
    return dict(mel=batch["mel"][idx][:, :, :T].contiguous(), text=batch["text"][idx][:, :Lk].contiguous(),
                speaker_ids=batch["speaker_ids"][idx], in_lens=in_lens, out_lens=out_lens,
                gate=batch["gate"][idx][:, :T].contiguous(), prior=beta_binomial_prior_batch(in_lens, out_lens, T, Lk))


def cpu_baseline_worker(batch_size, seed, hip_path=None):
    """The CPU oracle (oracle/flowtron_oracle.py = restatement of the reference, pinned to golden vectors made
    with the real reference) on a BOUNDED sample: the CPU_SAMPLE_UTTS shortest utterances of rank 0's batch (~10 s of CPU work
    on the GPU box's 16 usable cores), full forward + loss + backward, fp32, all usable host cores, torch's CPU LSTM over the padded
    batch (O.lstm_seq_padded: ~10x faster than the packed CPU path the reference itself would take).  Runs in its own process (no GPU context).  When the main
    process saved the HIP path's losses and gradients for the same sample and the same weights (hip_path), the worker also
    returns the parity figures of the benchmarked dtype against the oracle."""
    from oracle import flowtron_oracle as O
    import flowtron
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    model = flowtron.Flowtron(**MODEL_CONFIG)
    init_weights(model, 1234)
    batch = synth_batch(batch_size, seed)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    n_utt = CPU_SAMPLE_UTTS
    smp = sample_of(batch, n_utt)
    out_lens, in_lens, mel, text, pr, gate = smp["out_lens"], smp["in_lens"], smp["mel"], smp["text"], smp["prior"], smp["gate"]
    idx = slice(None)
    batch = dict(batch, speaker_ids=smp["speaker_ids"])
    T, Lk = mel.shape[2], text.shape[1]
    O.LSTM_IMPL["fn"] = O.lstm_seq_padded    # torch's CPU LSTM over the padded batch: several times faster than the packed path the
    best = None                               # reference takes on CPU (its autograd zero-fills [sum(lens), 4H] once per step)
    for it in range(2):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.forward(sd, MODEL_CONFIG, mel, batch["speaker_ids"][idx], text, in_lens, out_lens, pr)
        nll, gl, ctc = O.loss(out, gate, in_lens, out_lens, 1.0, True, True, -8)
        (nll + gl + 0.01 * ctc).sum().backward()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if dt > 20.0:                       # slow host: one pass is already a sample of the intended size
            break
    frames = int(out_lens.sum())
    res = {"value": round(frames / best, 2), "unit": "mel-frames/s", "cores": cores, "kind": "port",
           "sample": "%d shortest utterances of the batch (%d valid frames, T=%d, L=%d), fwd+loss+bwd, fp32, best of <=2, %.2f s"
                     % (n_utt, frames, T, Lk, best)}
    if hip_path and os.path.exists(hip_path):
        hip = torch.load(hip_path, weights_only=False)
        worst, worst_wc = ("", 0.0), ("", 0.0)
        for k, v in sd.items():
            r = v.grad
            if k.startswith("encoder.convolutions") and k.endswith("conv.bias"):
                continue                              # mathematically zero (instance norm removes the bias): |g| ~ 1e-10
            e = (hip["grads"][k] - r).norm().item() / max(r.norm().item(), 1e-30)
            if e > worst[1]:
                worst = (k, e)
            if not any(t in k for t in ILL_CONDITIONED) and e > worst_wc[1]:
                worst_wc = (k, e)
        hn, hg, hc = hip["losses"]
        res["parity"] = {
            "against": "CPU oracle (fp32) on the same %d utterances and the same weights; HIP side in the benchmarked dtype (%s operands)" % (n_utt, hip["dtype"]),
            "nll_rel": round(abs(hn - nll.item()) / abs(nll.item()), 6), "gate_abs": round(abs(hg - gl.item()), 6),
            "ctc_rel": round(abs(hc - ctc.item()) / max(abs(ctc.item()), 1e-30), 6),
            "worst_grad_rel": round(worst[1], 5), "worst_grad_name": worst[0],
            "worst_grad_rel_well_conditioned": round(worst_wc[1], 5), "worst_grad_name_well_conditioned": worst_wc[0],
            "sample": "the %d shortest utterances of the timed batch (T <= %d of 862 frames: what the CPU oracle finishes in seconds)" % (n_utt, int(out_lens.max())),
            "full_batch": full_batch_parity_from_log(),
            "note": "relative L2 per parameter tensor.  Round 5: the encoder convolutions' forward products come from split (hi|lo|hi) images "
                    "at fp32 grade -- the 16-bit rounding of those three GEMMs was what made embedding.weight / the encoder deviate 0.11 "
                    "(0.13-0.18 for the REAL reference under bf16 autocast, tests/golden/cfg2_bf16.pt); FLOWTRON_ENCODER_F32=off restores it"}
    return res


def stft_block(batch=32, seconds=10.0):
    """The mel front end (audio_processing.TacotronSTFT.mel_spectrogram -> ft_stft_r8: rFFT-1024 as a 512-point complex radix-8 FFT +
    triangular filterbank + log, one wave per frame) on a batch of LJS-shaped audio, against BOTH roofs: 1 344
    algorithmic HBM bytes per frame (1 024 B of new samples at hop 256 x 4 B + 320 B of mel out) and ~51 kFLOP of fp32 vector math per
    frame (5 N log2 N for the 512-point complex FFT = 23 k, the real-FFT split 4 k, |.|, 513 x <= 2 filterbank MACs, windowing, log:
    ~38 FLOP per byte) -- at 8 TB/s the arithmetic would need 300 TFLOP/s of fp32 VALU, twice the 157 TF peak, so the VALU roof is
    the nearer one; the kernel itself is LDS-transpose- and issue-bound below both."""
    import audio_processing
    stft = audio_processing.TacotronSTFT(1024, HOP, 1024, 80, SR, 0.0, 8000.0)
    y = torch.rand(batch, int(seconds * SR), device="cuda") * 2 - 1
    stft.mel_spectrogram(y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mel = stft.mel_spectrogram(y)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[len(ts) // 2]
    frames = int(mel.shape[0] * mel.shape[2])
    fps = frames / (ms * 1e-3)
    flop_per_frame, bytes_per_frame = 51e3, 1344
    return {"kernel": "stft_r8_k", "workload": "%d utterances x %.0f s at 22 050 Hz, hop 256, 1024-point rFFT, 80 mel bins" % (batch, seconds),
            "ms_per_batch": round(ms, 4), "frames": frames, "mframes_per_s": round(fps / 1e6, 1),
            "hbm": {"achieved": round(fps * bytes_per_frame / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(fps * bytes_per_frame / 8e12, 4),
                    "bytes_per_frame": bytes_per_frame},
            "valu_fp32": {"achieved": round(fps * flop_per_frame / 1e12, 2), "peak": 157.0, "unit": "TFLOP/s", "frac": round(fps * flop_per_frame / 157e12, 4),
                          "flop_per_frame": flop_per_frame},
            "bound": "valu_fp32 (38 FLOP per algorithmic byte: the vector-math roof is the nearer one; the kernel is LDS-transpose / issue bound below it)"}


def full_batch_parity_from_log():
    """the figures tests/test_gpu_bench_path.py::test_bf16_benchmark_config_at_its_own_shape_vs_oracle printed in the newest committed
    run of the GPU suite (profiles/*pytest_gpu*.log): the timed batch itself -- B 32, T 862, L 157, 18 932 valid frames -- against the
    fp32 oracle.  Parsed, not typed in; null fields when no log carries the block."""
    import glob
    import re
    out = {"test": "tests/test_gpu_bench_path.py::test_bf16_benchmark_config_at_its_own_shape_vs_oracle (-m gpu, run with -s / -rP): the "
                   "timed batch itself against the fp32 oracle", "last_run": None}
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pytest_gpu*.log")), key=lambda q: (re.findall(r"r(\d+)", os.path.basename(q)) or ["0"])[0] + os.path.basename(q))
    for path in reversed(logs):
        txt = open(path, errors="replace").read()
        m = re.search(r"\[bf16 config\[1\] at B 32 / T 862 vs fp32 oracle\] nll ([0-9.]+) / ([0-9.]+)\s+gate ([0-9.]+) / ([0-9.]+)\s+ctc ([0-9.]+) / ([0-9.]+)", txt)
        if not m:
            continue
        v = [float(x) for x in m.groups()]
        rows = re.findall(r"^\s+(\S+)\s+rel-L2 ([0-9.]+) \(tol ([0-9.]+); the reference's own bf16 run at T 862: ([0-9.]+)\)", txt[m.end():], re.M)
        out.update(last_run=os.path.relpath(path, ROOT), nll_rel=round(abs(v[0] - v[1]) / abs(v[1]), 8), gate_abs=round(abs(v[2] - v[3]), 6),
                   ctc_rel=round(abs(v[4] - v[5]) / max(abs(v[5]), 1e-30), 8))
        if rows:
            worst = max(rows, key=lambda r: float(r[1]))
            out.update(worst_grad_rel=float(worst[1]), worst_grad_name=worst[0], worst_grad_tol=float(worst[2]),
                       real_reference_under_bf16_autocast_same_tensor=float(worst[3]))
        break
    return out


def cpu_baseline(batch_size, seed, timeout_s=420, hip_path=None):
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(batch_size), str(seed)]
                           + ([hip_path] if hip_path else []),
                           capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "cpu oracle sample did not finish in %d s" % timeout_s}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": "worker failed: " + (r.stderr or "")[-400:]}


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


class _SyntheticAudioSet(torch.utils.data.Dataset):
    """The bench batch as a DATASET the drop-in loader path consumes: item i = (AudioItem of synthetic audio whose frame count is
    utterance i's out_len, speaker id, text ids, None) -- what flowtron_amd.data.Data.__getitem__ yields (reference data.py:173-186),
    minus the file read and the text front end."""

    def __init__(self, batch_cpu, steps):
        from flowtron_amd.data import AudioItem
        self.b, self.steps, self.AudioItem = batch_cpu, steps, AudioItem
        self.stft_args = dict(filter_length=1024, hop_length=HOP, win_length=1024, n_mel_channels=80, sampling_rate=SR, mel_fmin=0.0,
                              mel_fmax=8000.0)
        rs = np.random.RandomState(7)
        self.audio = [torch.from_numpy((0.1 * rs.standard_normal((int(t) - 1) * HOP + 17)).astype(np.float32)) for t in batch_cpu["out_lens"]]

    def __len__(self):
        return len(self.audio) * self.steps

    def __getitem__(self, i):
        j = i % len(self.audio)
        a = self.audio[j]
        return (self.AudioItem(a, a.numel() // HOP + 1, self.stft_args, (1.0, 0.0)), self.b["speaker_ids"][j:j + 1],
                self.b["text"][j, :int(self.b["in_lens"][j])], None)


def trainpy_step_block(model, criterion, optimizer, batch_cpu, steps, warmup, use_prior):
    """The step as the reference's UNMODIFIED train.py:282-331 executes it on the drop-in modules, timed beside
    the headline: batches from a torch DataLoader (train.py:74-80: one worker, batch_size, drop_last) through
    flowtron_amd.data.DataCollate, the seven `.cuda()` calls (the mel and prior slots are DeferredMel / DeferredPrior: rFFT + mel
    filterbank and the beta-binomial prior run on the device there), model.zero_grad(), autocast(enabled=False), the four
    loss `.item()` host reads, backward, torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0), optimizer.step().
    Same model, optimizer, batch shape and lengths as the headline; the mel VALUES come from synthetic audio."""
    from flowtron_amd.data import DataCollate
    ds = _SyntheticAudioSet(batch_cpu, steps + warmup)
    B = len(ds.audio)
    loader = torch.utils.data.DataLoader(ds, num_workers=1, shuffle=False, sampler=None, batch_size=B, pin_memory=False, drop_last=True,
                                         collate_fn=DataCollate(1, use_prior))
    ctc_w = criterion.ctc_loss_weight
    t_data, t_step, frames, host = [], [], 0, {}
    it = iter(loader)
    model.train()
    for i in range(steps + warmup):
        batch = next(it)                                  # the worker prefetches: host-side collate overlaps the previous step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.zero_grad()
        (mel, spk_ids, txt, in_lens, out_lens, gate_target, attn_prior) = batch
        mel, spk_ids, txt = mel.cuda(), spk_ids.cuda(), txt.cuda()
        in_lens, out_lens = in_lens.cuda(), out_lens.cuda()
        gate_target = gate_target.cuda()
        attn_prior = attn_prior.cuda() if attn_prior is not None else None
        torch.cuda.synchronize()                          # (measurement only: splits data time from step time)
        t1 = time.perf_counter()
        with torch.amp.autocast("cuda", enabled=False):
            out = model(mel, spk_ids, txt, in_lens, out_lens, attn_prior)
            loss_nll, loss_gate, loss_ctc = criterion(out, gate_target, in_lens, out_lens, is_validation=False)
            loss = loss_nll + loss_gate
            loss += loss_ctc * ctc_w
        ta = time.perf_counter()                          # forward + loss enqueued (host time only)
        reduced = (loss.item(), loss_gate.item(), loss_nll.item(), loss_ctc.item())
        tb = time.perf_counter()                          # ... and executed: the four host reads of train.py:307-321
        loss.backward()
        tc = time.perf_counter()                          # backward enqueued
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        td = time.perf_counter()
        optimizer.step()
        te = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= warmup:
            t_data.append(t1 - t0)
            t_step.append(t2 - t1)
            for key, val in (("fwd_enqueue", ta - t1), ("items_wait", tb - ta), ("bwd_enqueue", tc - tb), ("torch_clip_enqueue", td - tc),
                             ("optimizer_step_enqueue", te - td), ("final_sync", t2 - te)):
                host[key] = host.get(key, 0.0) + val
            frames += int(out_lens.sum().item())
    del it
    ms = sum(t_step) / len(t_step) * 1e3
    dms = sum(t_data) / len(t_data) * 1e3
    return {"what": "train.py:282-331 call for call on the drop-in modules: DataLoader(num_workers=1) -> DataCollate -> 7x .cuda() "
                    "(DeferredMel: ft_stft_r8 + mel on device; DeferredPrior: ft_beta_binomial_prior) -> model.zero_grad() -> forward + loss -> "
                    "4x .item() -> backward -> torch.nn.utils.clip_grad_norm_ -> optimizer.step()",
            "steps": len(t_step), "ms_per_step": round(ms, 2), "data_ms_per_batch": round(dms, 2),
            "ms_per_step_incl_data": round(ms + dms, 2), "mel_frames_per_s_incl_data": round(frames / (sum(t_step) + sum(t_data)), 1),
            "host_ms": {k: round(v / len(t_step) * 1e3, 2) for k, v in host.items()},
            "final_loss": round(reduced[0], 5), "T_max": int(mel.shape[2]), "L_max": int(txt.shape[1])}


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        print(json.dumps(cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100: a ~4 s timed region; 3 for --config ljs_cumm)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--mfma", default=None, choices=["bf16", "f16", "f32"], help="MFMA operand type (default: the config's)")
    ap.add_argument("--config", default="ljs", choices=["ljs", "libritts", "libritts_fp16", "ljs_cumm"],
                    help="ljs = BASELINE configs[1] (the headline line); libritts = configs[2] (123 speakers, L <= 237, bf16); "
                         "libritts_fp16 = configs[4] (fp16 operands + GradScaler, no attention prior); ljs_cumm = configs[1] with "
                         "use_cumm_attention (location-sensitive attention, SURVEY 8a row a17: the key projection is redone every frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--hidden", type=int, default=None,
                    help="n_hidden of the attention / decoder LSTMs (default: the config's 1024).  A variant of the workload, not the "
                         "BASELINE line: the roofline / parity / infer / trainpy legs are skipped")
    ap.add_argument("--no-pad-hidden", action="store_true",
                    help="with --hidden < 1024: the launch-per-step recurrence kernels instead of the zero-padded persistent ones (yardstick)")
    ap.add_argument("--no-trainpy", action="store_true", help="skip the train.py-call-sequence block (trainpy_step)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs an MI355X: the product path has no CPU fallback"
    if args.steps is None:
        args.steps = 3 if args.config == "ljs_cumm" else 100
    if args.mfma is None:
        args.mfma = "f16" if args.config == "libritts_fp16" else "bf16"
    os.environ["FLOWTRON_MFMA"] = args.mfma
    libri = args.config in ("libritts", "libritts_fp16")
    use_prior = args.config != "libritts_fp16"
    model_config = dict(MODEL_CONFIG, n_speakers=123) if libri else dict(MODEL_CONFIG, use_cumm_attention=args.config == "ljs_cumm")
    if args.hidden is not None and args.hidden != MODEL_CONFIG["n_hidden"]:
        model_config["n_hidden"] = args.hidden
        args.no_infer = args.no_trainpy = args.no_cpu_baseline = True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "6000")
    # test hook (tests/test_gpu_dist.py): the N > 1 control flow of this script on a ONE-GPU box -- every rank on cuda:0 and gloo
    # instead of RCCL (which refuses two ranks on one device).  Never set by the driver; the line then says so in `config`.
    shared_gpu = os.environ.get("BENCH_SHARED_GPU", "0") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    import flowtron
    from flowtron_amd import _lib as L
    from flowtron_amd import dist as ftdist
    from flowtron_amd.optim import RAdam
    import torch.distributed as dist
    L.lib()                                                   # fail loudly if the HIP library is missing
    if args.no_pad_hidden:
        from flowtron_amd import ops as _ops0
        _ops0._PAD_H = False
    if world > 1:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # the reference-style "> initializing distributed" chatter
            os.environ["LOCAL_RANK"] = str(local_rank)
            ftdist.init_distributed(rank, world, "gloo" if shared_gpu else "nccl", None)
        assert dist.get_world_size() == args.gpus, "process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus)

    torch.manual_seed(1234)
    model = flowtron.Flowtron(**model_config)
    init_weights(model, 1234)
    model = model.cuda().train()
    criterion = flowtron.FlowtronLoss(sigma=1.0, gm_loss=False, gate_loss=True, use_ctc_loss=True, ctc_loss_weight=0.01,
                                      blank_logprob=-8)
    optimizer = RAdam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    if world > 1:
        model = ftdist.apply_gradient_allreduce(model)
        model._comm_timing = True                                 # HIP events around the end-of-backward exchange (dist.py)

    if libri:
        batch_cpu = synth_batch(args.batch, 1234 + 7 + rank, l_max=237, l_min=5, n_speakers=123, chars_per_frame=1 / 3.6)
    else:
        batch_cpu = synth_batch(args.batch, 1234 + 7 + rank)
    T, Lk = batch_cpu["mel"].shape[2], batch_cpu["text"].shape[1]
    b = {k: v.cuda() for k, v in batch_cpu.items()}
    prior = beta_binomial_prior_batch(batch_cpu["in_lens"], batch_cpu["out_lens"], T, Lk).cuda() if use_prior else None
    frames_rank = int(batch_cpu["out_lens"].sum())
    # fp16 operands: the reference's AMP step (train.py:292-331) -- GradScaler around backward, unscale_, clip, scaler.step
    scaler = torch.amp.GradScaler("cuda", enabled=args.mfma == "f16")

    def step():
        optimizer.zero_grad()
        out = model(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], prior)
        nll, gl, ctc = criterion(out, b["gate"], b["in_lens"], b["out_lens"])
        loss = nll + gl + criterion.ctc_loss_weight * ctc
        if scaler.is_enabled():
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
            optimizer.clip_grad_norm_(1.0)
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.clip_grad_norm_(1.0)
            optimizer.step()
        return loss

    hip_path = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "ljs":
        # parity of the benchmarked dtype: the HIP path on the oracle's bounded sample, with the initial weights (before any
        # update), dropout off like the oracle; the cpu_baseline worker compares (it holds the oracle's gradients)
        try:
            smp = {k: v.cuda() for k, v in sample_of(batch_cpu).items()}
            model.eval()
            optimizer.zero_grad()
            out = model(smp["mel"], smp["speaker_ids"], smp["text"], smp["in_lens"], smp["out_lens"], smp["prior"])
            nll, gl, ctc = criterion(out, smp["gate"], smp["in_lens"], smp["out_lens"])
            (nll + gl + criterion.ctc_loss_weight * ctc).backward()
            torch.cuda.synchronize()
            hip_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "bench_hip_parity_%d.pt" % os.getpid())
            torch.save({"losses": (nll.item(), gl.item(), ctc.item()), "dtype": args.mfma,
                        "grads": {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}}, hip_path)
            del out, smp
        except Exception as e:
            log("parity pass failed: %r" % (e,))
            hip_path = None
        model.train()
        optimizer.zero_grad()

    log("model + batch ready (%d valid frames/step); warm-up" % frames_rank)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if os.environ.get("BENCH_HOST_PROFILE"):          # debugging aid: where does the HOST spend its time enqueueing a step?
        import cProfile, io, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            step()
        pr.disable()
        torch.cuda.synchronize()
        sio = io.StringIO()
        pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(60)
        log(sio.getvalue())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log("timed region done: %.2f ms/step" % (dt / max(args.steps, 1) * 1e3))
    comm_ms = None
    if world > 1 and getattr(model, "_comm_events", None):
        evs = model._comm_events[-args.steps:]
        comm_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        model._comm_timing = False
    from flowtron_amd import ops as _ops_chk
    _ops_chk.check_persist_status()                              # a persistent recurrence that timed out would have produced garbage
    loss_val = float(loss.item())
    stats = torch.tensor([dt, float(frames_rank)], dtype=torch.float64, device="cuda")
    comm_all = None
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fsum = stats[1:].clone()
        dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
        dt, frames_all = float(tmax.item()), float(fsum.item())
        cm = torch.zeros(world, dtype=torch.float64, device="cuda")
        cm[rank] = comm_ms if comm_ms is not None else -1.0
        dist.all_reduce(cm, op=dist.ReduceOp.SUM)
        comm_all = [round(float(v), 3) for v in cm.tolist()]
    else:
        frames_all = float(frames_rank)
    skipped = int(optimizer.skipped_steps)

    if rank == 0:
        res = {
            "metric": "mel-frames/sec training (2-flow, 80-mel)", "value": round(frames_all * args.steps / dt, 1),
            "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
            "config": {"workload": "%s, per-GPU batch %d, T_max=%d, L_max=%d, %s + CTC on, fwd+loss+bwd+%sclip+RAdam%s"
                                   % ({"ljs": "BASELINE configs[1]: 2-flow LJS config.json model",
                                       "ljs_cumm": "BASELINE configs[1] model with use_cumm_attention=True (location-sensitive attention)",
                                       "libritts": "BASELINE configs[2]: 2-flow LibriTTS model (123 speakers)",
                                       "libritts_fp16": "BASELINE configs[4]: 2-flow LibriTTS model (123 speakers), fp16 operands + GradScaler"}[args.config],
                                      args.batch, T, Lk, "attn-prior" if use_prior else "no attn-prior",
                                      "unscale+" if scaler.is_enabled() else "", (", %s RCCL all-reduce(AVG) of the gradient arena %s" % (("per-flow bucketed", "under backward") if getattr(model, "_grad_overlap", False) else ("ONE in-place", "at the end of backward"))) if world > 1 else ""),
                       "global_batch": args.batch * world, "valid_frames_per_step": int(frames_all),
                       "padded_frames_per_step": args.batch * T * world, "parallelism": "dp%d" % world,
                       "mfma_operands": args.mfma, "storage": "fp32", "final_loss": round(loss_val, 5),
                       **({"n_hidden": model_config["n_hidden"], "variant": "NOT the BASELINE model: n_hidden %d instead of 1024 (%s recurrence kernels)"
                           % (model_config["n_hidden"], "launch-per-step" if args.no_pad_hidden else "zero-padded persistent")}
                          if model_config["n_hidden"] != MODEL_CONFIG["n_hidden"] else {}),
                       **({"test_hook": "BENCH_SHARED_GPU: all ranks on ONE GPU over gloo -- not a scaling measurement"} if shared_gpu else {})},
        }
        res["config"]["skipped_steps"] = skipped             # updates dropped by the device-side non-finite-norm guard (0 = none)
        n1_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "flowtron_bench_n1_%s.json" % args.config)
        if world == 1:
            try:
                json.dump({"value": res["value"], "ms_per_step": res["ms_per_step"], "steps": args.steps}, open(n1_path, "w"))
            except OSError:
                pass
        else:
            # self-diagnosing N > 1 line: what the collective cost on every rank, on which backend, beside the N = 1
            # run of the same box when the driver ran it first (SCALE runs N = 1, 2, 4, 8 back to back)
            n1 = None
            try:
                n1 = json.load(open(n1_path))
            except (OSError, ValueError):
                pass
            arena_mb = model._grad_arena.numel * 4 / 1e6
            res["dp"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                         "collectives_per_step": len(model._grad_buckets), "regime": "overlap" if model._grad_overlap else "end-of-backward",
                         "arena_mb": round(arena_mb, 1), "allreduce_ms_per_rank": comm_all,
                         "exposed_comm_ms": None if not comm_all else round(max(comm_all), 3),
                         "allreduce_busbw_gb_s": None if not comm_all or max(comm_all) <= 0 else
                         round(2 * (world - 1) / world * arena_mb / 1e3 / (max(comm_all) * 1e-3), 1),
                         "n1_same_box": n1,
                         "weak_scaling_efficiency_vs_n1": None if not n1 else round(res["value"] / (world * n1["value"]), 4),
                         "design_prediction": {"2": 0.91, "4": 0.94, "8": 0.96}.get(str(world)),

                         "note": "allreduce_ms = HIP events on the compute stream around the end-of-backward exchange (poison check + "
                                 "collective(s) + stream wait); with the end-of-backward regime all of it is exposed"}
        mode = {"bf16": L.FT_BF16, "f16": L.FT_F16, "f32": L.FT_F32}[args.mfma]
        if world == 1 and not args.no_trainpy and args.config in ("ljs", "libritts"):
            try:
                log("train.py call sequence (DataLoader -> .cuda() -> zero_grad -> fwd -> 4x item -> bwd -> torch clip -> step) ...")
                res["trainpy_step"] = trainpy_step_block(model, criterion, optimizer, batch_cpu, min(args.steps, 20), 3, use_prior)
                res["trainpy_step"]["gap_to_headline_ms"] = round(res["trainpy_step"]["ms_per_step"] - res["ms_per_step"], 2)
            except Exception as e:
                res["trainpy_step"] = {"error": repr(e)}
        log("roofline kernel timing ...")
        # Top level: the WHOLE STEP against the MFMA roof: valid frames/s x 325 MFLOP of algorithmic work per valid frame
        # (2 flows, forward + backward, padding and the tanh recomputation not counted) over the dense bf16 peak.  The step is a chain
        # of latency-bound recurrences with GEMMs in between, so the dominant KERNELS are reported under their own bound below it.
        frames_per_gpu = res["value"] / world
        res["roofline"] = {"bound": "mfma", "scope": "whole training step", "achieved": round(frames_per_gpu * 325e6 / 1e12, 2),
                           "peak": 2500.0, "unit": "TFLOP/s", "frac": round(frames_per_gpu * 325e6 / 2.5e15, 5), "traffic": None,
                           "flop_per_valid_frame": 325e6,
                           "gemm_mfma_busy_frac_pmc": {k: dict(zip(("frac", "pmc_round"), pmc_value("MFMA_BUSY", k, "mfma_busy_frac", with_source=True)))
                                                       for k in ("gemm_bf16_k<true, true, true, 256, 32, false>", "gemm_bf16_k<false, true, false, 128, 64, true>",
                                                                 "gemm_bf16_k<false, false, false, 128, 64, true>",
                                                                 "gemm_bf16_k<false, false, false, 128, 32, false>")}}
        if args.config == "ljs_cumm":
            try:
                res["roofline"]["cumulative_attention"] = cumm_roofline(args.batch, batch_cpu["in_lens"], mode)
            except Exception as e:
                res["roofline"]["cumulative_attention"] = {"error": repr(e)}
        variant = model_config["n_hidden"] != MODEL_CONFIG["n_hidden"]
        if variant:           # (a workload variant: the 325 MFLOP per frame and the kernel legs below describe the BASELINE model)
            res["roofline"] = {"bound": "mfma", "achieved": None, "peak": 2500.0, "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "note": "--hidden variant: not priced (the roofline legs describe the n_hidden = 1024 model)"}
        try:
            if variant:
                raise RuntimeError("skipped for a --hidden variant")
            from flowtron_amd import ops as _ops
            _dev = torch.device("cuda", torch.cuda.current_device())
            _slices = _ops.lstm_persist_slices(args.batch, MODEL_CONFIG["n_hidden"], False, mode, _dev)
            if _ops.lstm_persist_groups(args.batch, MODEL_CONFIG["n_hidden"], False, mode, _dev) or _slices:
                # dominant kernels of the step: the persistent recurrences (backward: six launches per step; forward: the attention
                # LSTM's launch + the decoder pair's role pipeline per flow) -- for --batch > 32 the first slice of 32 rows is timed
                n_sl = (args.batch + 31) // 32 if _slices else 1      # (share estimate only: a wider batch runs 8 / 16 rows per group)
                dom, second = persist_roofline(min(args.batch, 32), MODEL_CONFIG["n_hidden"], T, batch_cpu["out_lens"][:32], mode)
                for blk in (dom, second):
                    blk["share_of_step"] = round(n_sl * blk["us_per_training_step"] * 1e-3 / res["ms_per_step"], 3)
                    if n_sl > 1:
                        blk["launches_per_sequence"] = n_sl
                if second["share_of_step"] > dom["share_of_step"]:       # dominant = the larger share of the step
                    dom, second = second, dom
                res["roofline"]["dominant_kernel"], res["roofline"]["second_kernel"] = dom, second
            elif _ops.lstm2_supported(args.batch, MODEL_CONFIG["n_hidden"], mode):
                res["roofline"]["dominant_kernel"] = lstm2_step_roofline(args.batch, MODEL_CONFIG["n_hidden"], T)
                res["roofline"]["second_kernel"] = lstm_step_roofline(args.batch, MODEL_CONFIG["n_hidden"], T, mode)
            else:
                res["roofline"]["dominant_kernel"] = lstm_step_roofline(args.batch, MODEL_CONFIG["n_hidden"], T, mode)
        except Exception as e:                      # never lose the headline number to the side measurement
            res["roofline"]["dominant_kernel"] = {"error": repr(e)}
        if world == 1 and not args.no_infer and args.config == "ljs":
            model.eval()
            n_frames = int(os.environ.get("BENCH_INFER_FRAMES", "400"))      # (a PMC pass may shorten it)
            z = torch.randn(1, 80, n_frames, device="cuda") * 0.5
            text = b["text"][:1, :69]
            spk = b["speaker_ids"][:1]
            n_fl = MODEL_CONFIG["n_flows"]
            hops = handoff_hops()

            def time_infer(operands):
                """median wall time of Flowtron.infer (400 frames, gate disabled) with the decoder of that operand mode"""
                os.environ["FLOWTRON_MFMA"] = operands
                try:
                    for _ in range(2):
                        model.infer(z, spk, text, gate_threshold=1.0)       # warm-up (+ hipGraph capture / weight images)
                    torch.cuda.synchronize()
                    tis = []
                    for _ in range(int(os.environ.get("BENCH_INFER_CALLS", "7"))):
                        t1 = time.perf_counter()
                        mel, _ = model.infer(z, spk, text, gate_threshold=1.0)
                        torch.cuda.synchronize()
                        tis.append(time.perf_counter() - t1)
                finally:
                    os.environ["FLOWTRON_MFMA"] = args.mfma
                ti = sorted(tis)[len(tis) // 2]
                nf = int(mel.shape[2])
                return nf, ti, {"frames": nf, "seconds": round(ti, 5), "seconds_min": round(min(tis), 5), "seconds_max": round(max(tis), 5),
                                "calls": len(tis), "frames_per_s": round(nf / ti, 1), "rtf": round(ti / (nf * HOP / SR), 5),
                                "us_per_frame_per_flow": round(ti / nf / n_fl * 1e6, 2)}
            try:
                log("inference RTF (16-bit weight images, persistent decode) ...")
                nf, ti, blk = time_infer(args.mfma)
                # dec_persist_k streams nothing per frame (the flow's weights are register-resident), so neither HBM nor MFMA bounds
                # it: a frame is 9 dependent stages, each ending in a hand-off of its output vector -- 5 chip-wide (h_att, h0, h1, u1,
                # u2: one fabric hop into every XCD + one L2 hop to every CU) and 4 XCD-local (query, scores, context, conv output)
                floor = 5 * (hops["cross_xcd_hop_us"] + hops["same_xcd_hop_us"]) + 4 * hops["same_xcd_hop_us"]
                per = ti / nf / n_fl * 1e6
                blk["roofline"] = {"bound": "handoff-latency", "stages_per_frame": 9, "floor_us_per_frame_per_flow": round(floor, 2),
                                   "us_per_frame_per_flow": round(per, 2), "frac": round(floor / per, 3),
                                   "note": "floor = 5 x (cross-XCD hop + same-XCD hop) + 4 x same-XCD hop from profiles/r02_handoff_hops.json; the "
                                           "GEMV arithmetic of a stage (VALU, 26.8 M weights over 1024 waves) and host-side launch / "
                                           "result copies are what frac leaves"}
                blk["config"] = "2-flow LJS, B=1, L=69, sigma=0.5, %s images of the weights (NARROWER than the reference's fp32 inference.py; see infer_fp32), gate disabled" % args.mfma
                res["infer"] = blk
            except Exception as e:
                res["infer"] = {"error": repr(e)}
            try:
                log("inference RTF (fp32 weights: the reference's precision) ...")
                nf, ti, blk = time_infer("f32")
                wbytes = 26838656 * 4                                   # weights a frame of one flow must read, fp32
                ach = n_fl * wbytes * nf / ti / 1e9
                persist32 = os.environ.get("FLOWTRON_DECODE_PERSIST", "1") != "0"
                persist32_flag = lambda: persist32
                per32 = ti / nf / n_fl * 1e6
                floor32 = 5 * (hops["cross_xcd_hop_us"] + hops["same_xcd_hop_us"]) + 4 * hops["same_xcd_hop_us"]
                #  with the persistent launch NOTHING of this streams per frame: `achieved` is an EQUIVALENT bandwidth -- what a
                # decoder that streamed the flow's fp32 weights every frame would have to sustain to be as fast -- not a measured one.
                # The bound that describes the kernel is the one `infer` uses: 9 dependent stages per frame.
                blk["roofline"] = {"bound": "handoff-latency" if persist32_flag() else "hbm",
                                   "stages_per_frame": 9, "floor_us_per_frame_per_flow": round(floor32, 2), "us_per_frame_per_flow": round(per32, 2),
                                   "frac": round(floor32 / per32, 3) if persist32_flag() else round(ach / 8000.0, 4),
                                   "equivalent_bandwidth": {"kind": "EQUIVALENT, not measured: algorithmic fp32 weight bytes per frame / frame time",
                                                            "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4)},
                                   "bytes_per_frame_per_flow": wbytes,
                                   "note": ("algorithmic weight bytes per frame over the frame time.  dec_persist_k<true>: one launch per flow; the "
                                            "recurrent / large LSTM matrices (16.8 M of the 26.8 M weights) never move -- they are register-resident "
                                            "for the whole utterance --, the other 40 MB per frame and flow (query rows and the 1x1 conv once per XCD) "
                                            "are re-read from the L2 / Infinity Cache under the stage hand-offs")
                                   if persist32 else
                                   "fp32 weights streamed every frame by the staged hipGraph chain (L2 / Infinity Cache resident after the first frame)"}
                blk["config"] = ("2-flow LJS, B=1, L=69, sigma=0.5, fp32 weights and arithmetic = the reference's inference.py:68-71 "
                                 "(no autocast), %s, gate disabled" % ("one persistent launch per flow (dec_persist_k<true>)" if persist32
                                                                        else "staged hipGraph decode chain"))
                if persist32:                                           # the staged chain it replaces, timed beside it
                    os.environ["FLOWTRON_DECODE_PERSIST"] = "0"
                    try:
                        _, ti_s, blk_s = time_infer("f32")
                        blk["replaces"] = {"decoder": "staged hipGraph chain (8 launches per frame)", "rtf": blk_s["rtf"],
                                           "us_per_frame_per_flow": blk_s["us_per_frame_per_flow"]}
                    finally:
                        os.environ["FLOWTRON_DECODE_PERSIST"] = "1"
                res["infer_fp32"] = blk
            except Exception as e:
                res["infer_fp32"] = {"error": repr(e)}
        if world == 1 and not args.no_infer and args.config == "ljs":
            try:
                res["stft"] = stft_block()
            except Exception as e:
                res["stft"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and args.config == "ljs":
            try:
                log("cpu baseline (oracle, bounded sample) ...")
                cb = cpu_baseline(args.batch, 1234 + 7 + rank, hip_path=hip_path)
                if "parity" in cb:
                    res["parity"] = cb.pop("parity")
                res["cpu_baseline"] = cb
            except Exception as e:
                res["cpu_baseline"] = {"error": repr(e)}
            finally:
                if hip_path and os.path.exists(hip_path):
                    os.remove(hip_path)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()                       # rank 0 is still timing its roofline kernels: everybody leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
