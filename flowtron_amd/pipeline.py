"""Stream-pipelined teacher-forced flow (training forward/backward of one AR_Step).

Why: the three recurrences of a flow (attention LSTM -> attention -> decoder LSTM layer 0 -> layer 1) are chains of
~860 dependent 5-7 us launches each; a chain is latency-bound (B = 32 rows per step) and leaves the chip idle, while
INDEPENDENT chains overlap almost perfectly (scripts/exp/concurrent_lstm.py).  The sequence is therefore
cut into time-chunks and the four stages

    A  input projection + attention-LSTM chunk           B  query projection, fused scores/softmax, context, gate
    C  layer-0 input projection + layer-0 LSTM chunk     D  layer-1 projection + LSTM chunk, dense x2, 1x1 conv, coupling

run on four HIP streams, chunk c of stage X waiting only for chunk c of stage X-1 (events), so that stage A works on
chunk c+3 while D finishes chunk c.  Each LSTM chunk is ONE hipGraph launch (ft_lstm_seq_*_range, use_graph) -- the
host could not feed three concurrent launch chains otherwise.  Backward needs no extra code: autograd replays every
node on the stream its forward ran on and synchronises producer/consumer streams, and a dummy `token` edge between
consecutive chunks of the same LSTM orders the chunk backward calls (the carries dh, dc live in the LSTM workspace).

All LSTM buffers (gx, y, saved gates/cell, dy, dgx, workspaces, lens) are persistent per (module, T, B): stable
addresses keep the hipGraph cache hot across training steps and remove allocator traffic from the chains.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from . import _lib as L
from . import ops


class LSTMState:
    """Persistent buffers of one recurrence at one (T, B, H)."""

    def __init__(self, T, B, H, device):
        f = dict(device=device, dtype=torch.float32)
        self.T, self.B, self.H = T, B, H
        self.gx = torch.empty(T, B, 4 * H, **f)
        self.y = torch.empty(T, B, H, **f)
        self.gates = torch.empty(T, B, 4 * H, **f)
        self.cell = torch.empty(T, B, H, **f)
        self.dy = torch.empty(T, B, H, **f)
        self.dgx = torch.empty(T, B, 4 * H, **f)
        nbytes = L.lib().ft_lstm_workspace_bytes(B, H)
        self.work_f = torch.empty(nbytes, device=device, dtype=torch.uint8)
        self.work_b = torch.empty(nbytes, device=device, dtype=torch.uint8)
        self.lens = torch.zeros(B, device=device, dtype=torch.int32)


_STATES: Dict[Tuple, LSTMState] = {}
_STREAMS: Dict[torch.device, list] = {}


def lstm_state(owner, role, T, B, H, device) -> LSTMState:
    key = (id(owner), role, T, B, H, str(device))
    st = _STATES.get(key)
    if st is None:
        st = _STATES[key] = LSTMState(T, B, H, device)
    return st


def stage_streams(device):
    s = _STREAMS.get(device)
    if s is None:
        s = _STREAMS[device] = [torch.cuda.Stream(device=device) for _ in range(4)]
    return s


def enabled(T: int) -> bool:
    v = os.environ.get("FLOWTRON_PIPELINE", "auto").lower()
    if v in ("0", "off", "false"):
        return False
    if v in ("1", "on", "true"):
        return T >= 2
    return False      # auto = off: measured (scripts/exp/pipe_timing*.py, profiles/) the chains do not overlap inside a training step yet


def chunk_len() -> int:
    return max(1, int(os.environ.get("FLOWTRON_CHUNK", "96")))


class LSTMChunkFn(torch.autograd.Function):
    """Steps [s0, s1) of a length-masked LSTM whose buffers live in `st`.  `token` chains consecutive chunks."""

    @staticmethod
    def forward(ctx, gx_c, w_hh, token, st, s0, s1, mode, use_graph):
        L.require_cuda(gx_c, w_hh)
        w_hh = ops._c(w_hh)
        st.gx[s0:s1].copy_(gx_c)
        T, B, H = st.T, st.B, st.H
        L.check(L.lib().ft_lstm_seq_fwd_range(L.ptr(st.gx), L.ptr(w_hh), L.ptr(st.lens), L.ptr(st.y), H, L.ptr(st.gates),
                                              L.ptr(st.cell), L.ptr(st.work_f), T, B, H, 0, mode, s0, s1, int(use_graph),
                                              L.stream()), "ft_lstm_seq_fwd_range")
        ctx.save_for_backward(w_hh)
        ctx.st, ctx.rng, ctx.mode, ctx.use_graph = st, (s0, s1), mode, use_graph
        return st.y[s0:s1], torch.zeros(1, device=gx_c.device)

    @staticmethod
    def backward(ctx, dy_c, dtoken):
        (w_hh,) = ctx.saved_tensors
        st, (s0, s1) = ctx.st, ctx.rng
        T, B, H = st.T, st.B, st.H
        st.dy[s0:s1].copy_(dy_c)
        L.check(L.lib().ft_lstm_seq_bwd_range(L.ptr(st.dy), H, L.ptr(w_hh), L.ptr(st.lens), L.ptr(st.gates), L.ptr(st.cell),
                                              L.ptr(st.dgx), L.ptr(st.work_b), T, B, H, 0, ctx.mode, s0, s1, int(ctx.use_graph),
                                              L.stream()), "ft_lstm_seq_bwd_range")
        dW = None
        if s0 == 0 and ctx.needs_input_grad[1]:          # last chunk of the backward sweep: dgx is complete for all T
            dW = torch.zeros_like(w_hh)
            if T > 1:
                ops.gemm_raw(st.dgx[1:], st.y[:-1], dW, 4 * H, H, (T - 1) * B, 1, 4 * H, H, 1, H, mode=ctx.mode, splitk=True)
        return st.dgx[s0:s1], dW, torch.zeros(1, device=dy_c.device), None, None, None, None, None


def ar_step_forward_pipelined(step, mel, text, in32, out32, attn_prior):
    """Pipelined equivalent of AR_Step.forward (model.py) for the default (non-cumulative) attention path."""
    T, B, M = mel.shape
    dev = mel.device
    mode = L.mfma_mode()
    use_graph = os.environ.get("FLOWTRON_LSTM_GRAPH", "1") != "0"
    CH = chunk_len()
    a, p, att = step.attention_lstm, step.lstm, step.attention_layer
    H = a.weight_hh_l0.shape[1]
    has_gate = hasattr(step, "gate_layer")
    main = torch.cuda.current_stream(dev)
    sA, sB, sC, sD = stage_streams(dev)

    # ---- shared, on the caller's stream
    mel0 = torch.cat([mel.new_zeros(1, B, M), mel[:-1]], 0)
    K = ops.linear(text, att.key.linear_layer.weight, None, mode=mode)
    V = ops.linear(text, att.value.linear_layer.weight, None, mode=mode)
    b_att = a.bias_ih_l0 + a.bias_hh_l0
    b_l0 = p.bias_ih_l0 + p.bias_hh_l0
    b_l1 = p.bias_ih_l1 + p.bias_hh_l1
    st_a, st_0, st_1 = (lstm_state(step, r, T, B, H, dev) for r in ("att", "l0", "l1"))
    for st in (st_a, st_0, st_1):
        st.lens.copy_(out32)
    conv_w = step.conv.weight.reshape(step.conv.weight.shape[0], -1)
    ev0 = torch.cuda.Event()
    ev0.record(main)
    for s in (sA, sB, sC, sD):
        s.wait_event(ev0)
    for t_, s_ in ((mel0, sA), (K, sB), (V, sB), (mel, sD), (b_att, sA), (b_l0, sC), (b_l1, sD)):
        t_.record_stream(s_)
    if attn_prior is not None:
        attn_prior.record_stream(sB)

    tokA = tokC = tokD = torch.zeros(1, device=dev)
    zs, outs, gates, attns, lps = [], [], [], [], []
    for s0 in range(0, T, CH):
        s1 = min(T, s0 + CH)
        with torch.cuda.stream(sA):
            gxa = ops.LinearFn.apply(a.weight_ih_l0, b_att, L.ACT_NONE, mode, mel0[s0:s1])
            h_att, tokA = LSTMChunkFn.apply(gxa, a.weight_hh_l0, tokA, st_a, s0, s1, mode, use_graph)
            evA = torch.cuda.Event()
            evA.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(evA)
            Q = ops.linear(h_att, att.query.linear_layer.weight, None, mode=mode)
            pr = attn_prior[:, s0:s1].contiguous() if attn_prior is not None else None
            attn_c, lp_c = ops.AttentionScoresFn.apply(Q, K, att.v.linear_layer.weight, in32, pr, att.temperature)
            ctx_c = ops.ContextFn.apply(attn_c, V, mode)
            if has_gate:
                g = step.gate_layer.linear_layer
                gate_c = ops.linear([h_att, ctx_c], g.weight, g.bias, mode=mode)
                gate_c.record_stream(main)
                gates.append(gate_c)
            ctx_c.record_stream(sC)
            attn_c.record_stream(main)
            lp_c.record_stream(main)
            evB = torch.cuda.Event()
            evB.record(sB)
        with torch.cuda.stream(sC):
            sC.wait_event(evB)
            gx0 = ops.LinearFn.apply(p.weight_ih_l0, b_l0, L.ACT_NONE, mode, h_att, ctx_c)
            h0, tokC = LSTMChunkFn.apply(gx0, p.weight_hh_l0, tokC, st_0, s0, s1, mode, use_graph)
            evC = torch.cuda.Event()
            evC.record(sC)
        with torch.cuda.stream(sD):
            sD.wait_event(evC)
            gx1 = ops.LinearFn.apply(p.weight_ih_l1, b_l1, L.ACT_NONE, mode, h0)
            h1, tokD = LSTMChunkFn.apply(gx1, p.weight_hh_l1, tokD, st_1, s0, s1, mode, use_graph)
            u = step.dense_layer(h1)
            out_c = ops.linear(u, conv_w, step.conv.bias, mode=mode)
            z_c = ops.AffineFn.apply(out_c, mel[s0:s1])
            out_c.record_stream(main)
            z_c.record_stream(main)
        zs.append(z_c)
        outs.append(out_c)
        attns.append(attn_c)
        lps.append(lp_c)
    for s in (sA, sB, sC, sD):
        main.wait_stream(s)
    z = torch.cat(zs, 0)
    out = torch.cat(outs, 0)
    log_s = out[..., :M]
    gate = torch.cat(gates, 0) if has_gate else None
    return z, log_s, gate, torch.cat(attns, 1), torch.cat(lps, 1)
