"""CPU oracle for the Flowtron hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (plain torch fp32 on CPU)
of the algorithm implemented by the reference NVIDIA/flowtron `flowtron.py`.
It exists so that the HIP kernels can be checked on a box where
`/root/reference` is not present.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it; the product package
`flowtron_amd` never does.

Parity pinning: `tests/golden/make_golden.py` imports the *real* reference
(`/root/reference/flowtron.py`, two shims) in the build container and writes
golden input/output vectors; `tests/test_oracle_golden.py` checks every
function here against those vectors (fp32, max-abs <= 2e-5 on activations).

All tensors use the reference's public layouts:
  mel [B,80,T], text int64 [B,L], in_lens/out_lens int64 [B],
  attn_prior [B,T,L]; returns z [T,B,80], log_s [T,B,80], gate [T,B,1],
  attn [B,T,L], attn_logprob [B,T,L].

Each function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def length_mask(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """bool [B,max_len], True where position < length (flowtron.py:39-50)."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(max_len, device=lengths.device)
    return ids[None, :] < lengths[:, None]


def reverse_by_length(x: torch.Tensor, lens: torch.Tensor, time_dim: int, batch_dim: int) -> torch.Tensor:
    """flip(time) followed by per-sample roll(+len) (flowtron.py:606-613).

    Closed form: y[t] = x[len-1-t] for t < len, y[t] = x[T-1+len-t] otherwise.
    The map is an involution, so it is also its own inverse (flowtron.py:619-626).
    """
    T = x.shape[time_dim]
    B = x.shape[batch_dim]
    t = torch.arange(T)[None, :]                      # [1,T]
    ln = lens.view(B, 1).to(torch.long)
    src = torch.where(t < ln, ln - 1 - t, T - 1 + ln - t)   # [B,T]
    xm = x.movedim((batch_dim, time_dim), (0, 1))           # [B,T,...]
    idx = src.view(B, T, *([1] * (xm.dim() - 2))).expand_as(xm)
    ym = torch.gather(xm, 1, idx)
    return ym.movedim((0, 1), (batch_dim, time_dim))


def lstm_cell_seq(x: torch.Tensor, lens: Optional[torch.Tensor], w_ih, w_hh, b_ih, b_hh,
                  reverse: bool = False, state=None, return_state: bool = False):
    """Length-masked single-layer LSTM, explicit recurrence (the definition).

    x [T,B,I] -> y [T,B,H]; y is zero at t >= len_b ("packed" semantics,
    flowtron.py:689-694).  Gate order i|f|g|o (torch.nn.LSTM layout).
    reverse=True runs each sample from len_b-1 down to 0 (BiLSTM reverse dir).
    """
    T, B, _ = x.shape
    H = w_hh.shape[1]
    if lens is None:
        lens = torch.full((B,), T, dtype=torch.long)
    h = x.new_zeros(B, H) if state is None else state[0].clone()
    c = x.new_zeros(B, H) if state is None else state[1].clone()
    gx = x @ w_ih.t() + (b_ih + b_hh)
    y = x.new_zeros(T, B, H)
    bidx = torch.arange(B)
    for s in range(T):
        active = (s < lens)
        t_b = torch.where(active, (lens - 1 - s) if reverse else torch.full_like(lens, s), torch.zeros_like(lens))
        a = gx[t_b, bidx] + h @ w_hh.t()
        i, f, g, o = a.chunk(4, dim=1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c_new = f * c + i * g
        h_new = o * torch.tanh(c_new)
        m = active[:, None]
        c = torch.where(m, c_new, c)
        h = torch.where(m, h_new, h)
        ya = torch.where(m, h_new, torch.zeros_like(h_new))
        y[t_b[active], bidx[active]] = ya[active]
    if return_state:
        return y, (h, c)
    return y


def lstm_seq_fast(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Same as lstm_cell_seq but through torch's fused CPU LSTM on a packed
    sequence -- exactly what the reference executes (flowtron.py:689-694).
    Used for the timed cpu_baseline and for large shapes."""
    T, B, _ = x.shape
    if lens is None:
        lens = torch.full((B,), T, dtype=torch.long)
    lens_s, order = torch.sort(lens.cpu(), descending=True)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(B)
    xs = x[:, order]
    if reverse:
        xs = reverse_valid(xs, lens_s)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xs, lens_s)
    out = torch._VF.lstm(packed.data, packed.batch_sizes, (x.new_zeros(1, B, w_hh.shape[1]),) * 2,
                         [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False)[0]
    out = torch.nn.utils.rnn.PackedSequence(out, packed.batch_sizes)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=T)
    if reverse:
        y = reverse_valid(y, lens_s)
    return y[:, inv]


def lstm_seq_padded(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Same function once more, for LONG sequences: torch's CPU LSTM over the PADDED batch, outputs of pad frames zeroed
    afterwards.  A forward-direction LSTM's output at t < len does not depend on later frames, so the valid outputs (and,
    through the mask, every gradient) equal those of the packed run of flowtron.py:689-694; the reversed direction runs on
    the per-sample reversed input like lstm_seq_fast.  Why it exists: autograd through the PACKED CPU LSTM narrows the
    [sum(lens), 4H] projection once per step, and every narrow's backward zero-fills a tensor of that full size -- O(T^2):
    at T = 862 that was 88 s of fill_ per four utterances.  Pinned to lstm_cell_seq in tests/test_oracle_golden.py."""
    T, B, _ = x.shape
    if lens is None:
        lens = torch.full((B,), T, dtype=torch.long)
    lens = lens.cpu().long()
    xs = reverse_valid(x, lens) if reverse else x
    h0 = x.new_zeros(1, B, w_hh.shape[1])
    y = torch._VF.lstm(xs, (h0, h0), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, True, False, False)[0]
    y = y * (torch.arange(T)[:, None] < lens[None, :]).unsqueeze(-1).to(y.dtype)
    return reverse_valid(y, lens) if reverse else y


def reverse_valid(x, lens):
    """Reverse each sample's valid span [0,len) in time (dim 0), pads untouched."""
    T, B = x.shape[:2]
    t = torch.arange(T)[:, None]
    ln = lens.view(1, B)
    src = torch.where(t < ln, ln - 1 - t, t)
    idx = src.view(T, B, *([1] * (x.dim() - 2))).expand_as(x)
    return torch.gather(x, 0, idx)


LSTM_IMPL = {"fn": lstm_cell_seq}


def _lstm(x, lens, sd, prefix, layer=0, reverse=False):
    sfx = "_l%d%s" % (layer, "_reverse" if reverse else "")
    return LSTM_IMPL["fn"](x, lens, sd[prefix + "weight_ih" + sfx], sd[prefix + "weight_hh" + sfx],
                           sd[prefix + "bias_ih" + sfx], sd[prefix + "bias_hh" + sfx], reverse=reverse)


# --------------------------------------------------------------------------
# encoder  (flowtron.py:467-525, 53-126)
# --------------------------------------------------------------------------
def masked_instance_norm(x, mask, weight, bias, eps=1e-5):
    """x [B,C,L], mask [B,1,L] float (or None). Biased variance over valid
    positions; applied to all positions (flowtron.py:73-90)."""
    if mask is None:
        n = x.shape[2]
        mean = x.mean(2, keepdim=True)
        var = ((x - mean) ** 2).mean(2, keepdim=True)
    else:
        n = mask.sum(2, keepdim=True)
        mean = (x * mask).sum(2, keepdim=True) / n
        var = (((x - mean) * mask) ** 2).sum(2, keepdim=True) / n
    return (x - mean) / torch.sqrt(var + eps) * weight[None, :, None] + bias[None, :, None]


def encoder(sd: SD, text_emb: torch.Tensor, in_lens: Optional[torch.Tensor],
            dropout_masks: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """text_emb [B,C,L] -> [B,L,C] (flowtron.py:492-514 forward; :516-525 infer
    when in_lens is None). dropout_masks: optional list of 3 keep-masks already
    scaled by 1/(1-p) (training-mode parity with injected masks)."""
    x = text_emb
    B = x.shape[0]
    use_mask = in_lens is not None and B > 1
    m = length_mask(in_lens, x.shape[2])[:, None, :].to(x.dtype) if use_mask else None
    for i in range(3):
        if m is not None:
            x = x * m                               # masked_fill_(~mask, 0) :501
        x = F.conv1d(x, sd["encoder.convolutions.%d.0.conv.weight" % i],
                     sd["encoder.convolutions.%d.0.conv.bias" % i], padding=2)
        x = masked_instance_norm(x, m, sd["encoder.convolutions.%d.1.weight" % i],
                                 sd["encoder.convolutions.%d.1.bias" % i])
        x = torch.relu(x)
        if dropout_masks is not None:
            x = x * dropout_masks[i]
    x = x.permute(2, 0, 1)                          # [L,B,C]
    lens = in_lens if in_lens is not None else None
    yf = _lstm(x, lens, sd, "encoder.lstm.", 0, reverse=False)
    yb = _lstm(x, lens, sd, "encoder.lstm.", 0, reverse=True)
    return torch.cat([yf, yb], 2).permute(1, 0, 2)  # [B,L,C]


# --------------------------------------------------------------------------
# attention (flowtron.py:528-592)
# --------------------------------------------------------------------------
def attention(sd: SD, pfx: str, queries, enc, pad_mask, attn_prior, temperature=1.0,
              key_scale=None, chunk: int = 64):
    """queries [T,B,H], enc [L,B,E], pad_mask bool [B,L] True=pad or None.
    Returns ctx [T,B,A], attn [B,T,L], attn_logprob [B,T,L]."""
    wq = sd[pfx + "query.linear_layer.weight"]
    wk = sd[pfx + "key.linear_layer.weight"]
    wv = sd[pfx + "value.linear_layer.weight"]
    v = sd[pfx + "v.linear_layer.weight"][0]
    kin = enc if key_scale is None else enc * key_scale
    K = (kin @ wk.t()).transpose(0, 1)                # [B,L,A]
    V = (enc @ wv.t()).transpose(0, 1)                # [B,L,A]
    Q = (queries @ wq.t()).transpose(0, 1)            # [B,T,A]
    B, T, _ = Q.shape
    e = Q.new_empty(B, T, K.shape[1])
    for t0 in range(0, T, chunk):                     # chunked: avoids the B*T*L*A tensor
        s = torch.tanh(Q[:, t0:t0 + chunk, None, :] + K[:, None, :, :])
        e[:, t0:t0 + chunk] = s @ v
    e = e / temperature
    if pad_mask is not None:
        e = e.masked_fill(pad_mask[:, None, :], -float("inf"))
    p = torch.softmax(e, dim=2)
    if attn_prior is not None:
        u = torch.log(p + 1e-20) + torch.log(attn_prior.float() + 1e-20)   # :546-548
        logprob = u.clone()
        if pad_mask is not None:
            u = u.masked_fill(pad_mask[:, None, :], -float("inf"))
        attn = torch.softmax(u, dim=2)
    else:
        attn = p
        logprob = torch.log(p + 1e-8)                 # :583
    ctx = torch.bmm(attn, V).transpose(0, 1)          # [T,B,A]
    return ctx, attn, logprob


# --------------------------------------------------------------------------
# one AR flow, teacher forced (flowtron.py:725-773)
# --------------------------------------------------------------------------
def attn_cond(sd: SD, pfx: str, cumm, prev):
    """AttentionConditioningLayer (flowtron.py:129-152): cumm, prev [B,1,L] -> key modulation [L,B,E]."""
    x = torch.cat([cumm, prev], 1)
    x = torch.relu(F.conv1d(x, sd[pfx + "location_conv_hidden.conv.weight"], sd[pfx + "location_conv_hidden.conv.bias"], padding=2))
    x = torch.sigmoid(F.conv1d(x, sd[pfx + "location_conv_out.conv.weight"], sd[pfx + "location_conv_out.conv.bias"], padding=1))
    return x.permute(2, 0, 1)


def cumm_attention_sequence(sd: SD, pfx: str, h_att, enc, pad_mask):
    """run_cumm_attn_sequence (flowtron.py:697-723): per-frame loop, keys modulated by the location features, values
    not; NB the reference drops the attention prior on this branch (:742-743)."""
    T, B, _ = h_att.shape
    Lk = enc.shape[0]
    cumm = enc.new_zeros(B, 1, Lk)
    prev = enc.new_zeros(B, 1, Lk)
    ctxs, attns, lps = [], [], []
    for i in range(T):
        cond = attn_cond(sd, pfx + "attn_cond_layer.", cumm, prev)
        ctx, prev, lp = attention(sd, pfx + "attention_layer.", h_att[i:i + 1], enc, pad_mask, None, key_scale=cond)
        ctxs.append(ctx); attns.append(prev); lps.append(lp)
        cumm = cumm + prev
    return torch.cat(ctxs, 0), torch.cat(attns, 1), torch.cat(lps, 1)


def ar_step_forward(sd: SD, pfx: str, mel, enc, pad_mask, out_lens, attn_prior, has_gate: bool):
    """mel [T,B,M] -> (z, log_s, gate|None, attn, attn_logprob)."""
    T, B, M = mel.shape
    mel0 = torch.cat([mel.new_zeros(1, B, M), mel[:-1]], 0)
    h_att = _lstm(mel0, out_lens, sd, pfx + "attention_lstm.", 0)
    if (pfx + "attn_cond_layer.location_conv_hidden.conv.weight") in sd:
        ctx, attn, logprob = cumm_attention_sequence(sd, pfx, h_att, enc, pad_mask)
    else:
        ctx, attn, logprob = attention(sd, pfx + "attention_layer.", h_att, enc, pad_mask, attn_prior)
    dec_in = torch.cat([h_att, ctx], 2)
    gate = None
    if has_gate:
        gate = dec_in @ sd[pfx + "gate_layer.linear_layer.weight"].t() + sd[pfx + "gate_layer.linear_layer.bias"]
    h = dec_in                                           # nn.LSTM(n_hidden + n_attn, n_hidden, n_lstm_layers), flowtron.py:655, :760-765
    layer = 0
    while (pfx + "lstm.weight_ih_l%d" % layer) in sd:
        h = _lstm(h, out_lens, sd, pfx + "lstm.", layer)
        layer += 1
    for i in range(2):
        h = torch.tanh(h @ sd[pfx + "dense_layer.layers.%d.linear_layer.weight" % i].t()
                       + sd[pfx + "dense_layer.layers.%d.linear_layer.bias" % i])
    out = h @ sd[pfx + "conv.weight"][:, :, 0].t() + sd[pfx + "conv.bias"]
    log_s, b = out[..., :M], out[..., M:]
    z = torch.exp(log_s) * mel + b
    return z, log_s, gate, attn, logprob


def flow_prefix(i: int) -> str:
    return "flows.%d." % i if i % 2 == 0 else "flows.%d.ar_step." % i


def embed_and_encode(sd: SD, speaker_ids, text, in_lens, dummy_speaker=False, dropout_masks=None):
    """(flowtron.py:872-887) -> enc [L,B,E]."""
    if dummy_speaker:
        speaker_ids = speaker_ids * 0
    spk = sd["speaker_embedding.weight"][speaker_ids.reshape(-1)]          # [B,S]
    emb = sd["embedding.weight"][text].transpose(1, 2)                     # [B,C,L]
    t = encoder(sd, emb, in_lens, dropout_masks).transpose(0, 1)          # [L,B,C]
    return torch.cat([t, spk[None].expand(t.shape[0], -1, -1)], 2)


def forward(sd: SD, cfg: dict, mel, speaker_ids, text, in_lens, out_lens, attn_prior=None,
            dropout_masks=None):
    """Flowtron.forward (flowtron.py:870-899). Returns the same 8-tuple."""
    n_flows = cfg["n_flows"]
    enc = embed_and_encode(sd, speaker_ids, text, in_lens, cfg.get("dummy_speaker_embedding", False),
                           dropout_masks)
    x = mel.permute(2, 0, 1)
    pad_mask = ~length_mask(in_lens, text.shape[1])
    log_s_list, attn_list, logprob_list = [], [], []
    gate = None
    for i in range(n_flows):
        has_gate = (i == n_flows - 1) and bool(cfg.get("use_gate_layer", True))
        pfx = flow_prefix(i)
        if i % 2 == 0:
            x, log_s, gate, attn, lp = ar_step_forward(sd, pfx, x, enc, pad_mask, out_lens, attn_prior, has_gate)
        else:
            xr = reverse_by_length(x, out_lens, 0, 1)
            pr = reverse_by_length(attn_prior, out_lens, 1, 0) if attn_prior is not None else None
            zr, log_s, gate, attn, lp = ar_step_forward(sd, pfx, xr, enc, pad_mask, out_lens, pr, has_gate)
            x = reverse_by_length(zr, out_lens, 0, 1)
        log_s_list.append(log_s)
        attn_list.append(attn)
        logprob_list.append(lp)
    return x, log_s_list, gate, attn_list, logprob_list, None, None, None


# --------------------------------------------------------------------------
# loss (flowtron.py:155-275)
# --------------------------------------------------------------------------
def attention_ctc_loss(attn_logprob, in_lens, out_lens, blank_logprob=-1.0):
    """attn_logprob [B,T,L] natural time order (flowtron.py:162-182)."""
    B = attn_logprob.shape[0]
    padded = F.pad(attn_logprob, (1, 0), value=blank_logprob)            # blank column first
    total = attn_logprob.new_zeros(())
    for b in range(B):
        K, Tq = int(in_lens[b]), int(out_lens[b])
        lp = F.log_softmax(padded[b, :Tq, :K + 1], dim=1)[:, None, :]    # [Tq,1,K+1]
        target = torch.arange(1, K + 1)[None]
        total = total + F.ctc_loss(lp, target, input_lengths=torch.tensor([Tq]),
                                   target_lengths=torch.tensor([K]), blank=0,
                                   reduction="mean", zero_infinity=True)
    return total / B


def loss(model_output, gate_target, in_lens, out_lens, sigma=1.0, gate_loss=True,
         use_ctc_loss=False, blank_logprob=-1.0):
    """FlowtronLoss.forward (flowtron.py:200-275), n_components == 0 branch."""
    z, log_s_list, gate_pred, attn_list, logprob_list = model_output[:5]
    T = z.shape[0]
    m = length_mask(out_lens, T).t()[..., None].to(z.dtype)              # [T,B,1]
    n = m.sum()
    log_s_total = sum((ls * m).sum() for ls in log_s_list)
    zz = z * m
    nll = ((zz * zz).sum() / (2 * sigma * sigma) - log_s_total) / (n * z.shape[2])
    gl = z.new_zeros(1)
    if gate_loss:
        gp = (gate_pred * m)[..., 0].t()                                # [B,T]
        g = F.binary_cross_entropy_with_logits(gp, gate_target, reduction="none")
        gl = (g.t() * m[:, :, 0]).sum() / n
    ctc = torch.zeros_like(gl)
    if use_ctc_loss:
        for i, lp in enumerate(logprob_list):
            if i % 2 == 1:
                lp = reverse_by_length(lp, out_lens, 1, 0)               # :250-256
            ctc = ctc + attention_ctc_loss(lp, in_lens, out_lens, blank_logprob)
        ctc = ctc / float(len(logprob_list))
    return nll, gl, ctc


# --------------------------------------------------------------------------
# inference (flowtron.py:775-828, 901-930)
# --------------------------------------------------------------------------
def _lstm_step(x, h, c, w_ih, w_hh, b_ih, b_hh):
    a = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = a.chunk(4, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def ar_step_infer(sd: SD, pfx: str, residual, enc, has_gate, temperature=1.0, gate_threshold=0.5,
                  attn_prior=None, attns=None, gates_out=None):
    """residual [N,1,M], enc [L,1,E] -> (mel [N',1,M], attn [N',L]) (flowtron.py:775-828).
    gates_out: optional list that receives (sigmoid(gate), gate input [h_att ; ctx]) of every decoded frame (the 400-frame decode
    test builds a gate with a wide stop margin from them)."""
    N, B, M = residual.shape
    H = sd[pfx + "lstm.weight_hh_l0"].shape[1]
    ap = pfx + "attention_layer."
    K = (enc @ sd[ap + "key.linear_layer.weight"].t()).transpose(0, 1)
    V = (enc @ sd[ap + "value.linear_layer.weight"].t()).transpose(0, 1)
    v = sd[ap + "v.linear_layer.weight"][0]
    z = residual.new_zeros
    ha, ca = z(B, H), z(B, H)
    n_layers = 0
    while (pfx + "lstm.weight_ih_l%d" % n_layers) in sd:
        n_layers += 1
    hs, cs = [z(B, H) for _ in range(n_layers)], [z(B, H) for _ in range(n_layers)]
    prev = z(B, M)
    outs, attn_rows = [], []
    cumm_on = (pfx + "attn_cond_layer.location_conv_hidden.conv.weight") in sd
    if cumm_on:
        cumm, prev_attn = z(B, 1, enc.shape[0]), z(B, 1, enc.shape[0])
        wk = sd[ap + "key.linear_layer.weight"]
    for i in range(N):
        ha, ca = _lstm_step(prev, ha, ca, sd[pfx + "attention_lstm.weight_ih_l0"], sd[pfx + "attention_lstm.weight_hh_l0"],
                            sd[pfx + "attention_lstm.bias_ih_l0"], sd[pfx + "attention_lstm.bias_hh_l0"])
        q = ha @ sd[ap + "query.linear_layer.weight"].t()                  # [B,A]
        if cumm_on:                                                        # flowtron.py:793-803
            K = ((enc * attn_cond(sd, pfx + "attn_cond_layer.", cumm, prev_attn)) @ wk.t()).transpose(0, 1)
        e = torch.tanh(q[:, None, :] + K) @ v / temperature               # [B,L]
        p = torch.softmax(e, dim=1)
        if attns is not None:                                              # forced alignment, flowtron.py:585-588
            p = attns[i].reshape(1, -1)
        elif attn_prior is not None:
            p = torch.softmax(torch.log(p + 1e-20) + torch.log(attn_prior[:, i].float() + 1e-20), dim=1)
        if cumm_on:
            prev_attn = p[:, None, :]
            cumm = cumm + prev_attn
        ctx = torch.bmm(p[:, None, :], V)[:, 0]                           # [B,A]
        d = torch.cat([ha, ctx], 1)
        u = d                                                              # nn.LSTM(.., n_lstm_layers) with carried (h, c), flowtron.py:654-655, :811-814
        for k in range(len(hs)):
            hs[k], cs[k] = _lstm_step(u, hs[k], cs[k], sd[pfx + "lstm.weight_ih_l%d" % k], sd[pfx + "lstm.weight_hh_l%d" % k],
                                      sd[pfx + "lstm.bias_ih_l%d" % k], sd[pfx + "lstm.bias_hh_l%d" % k])
            u = hs[k]
        for j in range(2):
            u = torch.tanh(u @ sd[pfx + "dense_layer.layers.%d.linear_layer.weight" % j].t()
                           + sd[pfx + "dense_layer.layers.%d.linear_layer.bias" % j])
        o = u @ sd[pfx + "conv.weight"][:, :, 0].t() + sd[pfx + "conv.bias"]
        log_s, b = o[:, :M], o[:, M:]
        prev = (residual[i] - b) / torch.exp(log_s)
        outs.append(prev)
        attn_rows.append(p[0] if B == 1 else p)                          # [L], or [B,L] for a batch (no gate: flowtron.py:823 needs B = 1)
        if has_gate:
            g = d @ sd[pfx + "gate_layer.linear_layer.weight"].t() + sd[pfx + "gate_layer.linear_layer.bias"]
            if gates_out is not None:
                gates_out.append((float(torch.sigmoid(g)), d[0].clone()))
            if float(torch.sigmoid(g)) > gate_threshold:
                break
    return torch.stack(outs, 0), torch.stack(attn_rows, 0)


def infer(sd: SD, cfg: dict, residual, speaker_ids, text, temperature=1.0, gate_threshold=0.5, attn_prior=None, attns=None,
          gates_out=None):
    """Flowtron.infer (flowtron.py:901-930). residual [1,M,N] -> (mel [1,M,N'], [attn per flow]).
    attn_prior [1,N,L]; attns: per flow (flows order) an [N,L] forced alignment in that flow's own time order."""
    n_flows = cfg["n_flows"]
    enc = embed_and_encode(sd, speaker_ids, text, None, cfg.get("dummy_speaker_embedding", False))
    x = residual.permute(2, 0, 1)
    attn_out = []
    for i in reversed(range(n_flows)):
        has_gate = (i == n_flows - 1) and bool(cfg.get("use_gate_layer", True))
        pfx = flow_prefix(i)
        fa = None if attns is None else attns[i]
        if i % 2 == 1:
            pr = None if attn_prior is None else torch.flip(attn_prior, (1,))
            xr, a = ar_step_infer(sd, pfx, torch.flip(x, (0,)), enc, has_gate, temperature, gate_threshold, pr, fa,
                                  gates_out if has_gate else None)
            x = torch.flip(xr, (0,))
        else:
            x, a = ar_step_infer(sd, pfx, x, enc, has_gate, temperature, gate_threshold, attn_prior, fa,
                                 gates_out if has_gate else None)
        attn_out.append(a)
    return x.permute(1, 2, 0), attn_out


# --------------------------------------------------------------------------
# audio front end (audio_processing.py:172-235, 96-134)
# --------------------------------------------------------------------------
def hann_periodic(n: int) -> torch.Tensor:
    """scipy.signal.get_window('hann', n, fftbins=True) (audio_processing.py:194)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * k / n))


def hz_to_mel_slaney(f):
    f = torch.as_tensor(f, dtype=torch.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return torch.where(f >= min_log_hz, min_log_mel + torch.log(torch.clamp(f, min=1e-10) / min_log_hz) / logstep, mel)


def mel_to_hz_slaney(m):
    m = torch.as_tensor(m, dtype=torch.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0) -> torch.Tensor:
    """Slaney-style mel filterbank == librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)
    with htk=False, norm='slaney' (called at audio_processing.py:104-105).
    librosa is a third-party dependency absent from /root/reference
    (requirements.txt:4 pins 0.6.3): PARITY UNPINNED for these constants --
    this restates the published Slaney formula."""
    fftfreqs = torch.linspace(0, sr / 2, 1 + n_fft // 2, dtype=torch.float64)
    mels = torch.linspace(float(hz_to_mel_slaney(fmin)), float(hz_to_mel_slaney(fmax)), n_mels + 2, dtype=torch.float64)
    mel_f = mel_to_hz_slaney(mels)
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = torch.clamp(torch.minimum(lower, upper), min=0)
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).float()


def stft_mel(y: torch.Tensor, n_fft=1024, hop=256, fb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """TacotronSTFT.mel_spectrogram (audio_processing.py:117-134, 207-235).
    y [B,N] in [-1,1] -> [B,80,N//hop+1]; DFT as a dense matrix product like the
    reference's conv1d basis."""
    if fb is None:
        fb = mel_filterbank(n_fft=n_fft)
    B, N = y.shape
    yp = F.pad(y[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    frames = yp.unfold(1, n_fft, hop)                                    # [B,F,n_fft]
    n = torch.arange(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * k[:, None] * n[None, :] / n_fft
    w = hann_periodic(n_fft)
    cosb = (torch.cos(ang) * w).float()
    sinb = (-torch.sin(ang) * w).float()
    re = frames @ cosb.t()
    im = frames @ sinb.t()
    mag = torch.sqrt(re * re + im * im)                                   # [B,F,513]
    mel = mag @ fb.t()
    return torch.log(torch.clamp(mel, min=1e-5)).transpose(1, 2)


# --------------------------------------------------------------------------
# beta-binomial attention prior (data.py:31-41)
# --------------------------------------------------------------------------
def beta_binomial_prior(P: int, M: int, scaling: float = 1.0) -> torch.Tensor:
    """[M,P] float64: pmf of BetaBinom(P-1, a=s*i, b=s*(M+1-i)) for i=1..M,
    closed form via lgamma."""
    k = torch.arange(P, dtype=torch.float64)[None, :]
    i = torch.arange(1, M + 1, dtype=torch.float64)[:, None]
    a, b = scaling * i, scaling * (M + 1 - i)
    n = float(P - 1)
    lg = torch.lgamma
    logc = lg(torch.tensor(n + 1, dtype=torch.float64)) - lg(k + 1) - lg(n - k + 1)
    logp = logc + lg(k + a) + lg(n - k + b) - lg(n + a + b) - (lg(a) + lg(b) - lg(a + b))
    return torch.exp(logp)


# --------------------------------------------------------------------------
# optimizer step of the caller row: RAdam as radam.py:44-120 states it, plus the global-norm clip of
# train.py:325-329 -- a functional restatement over a dict of tensors.  TEST INFRASTRUCTURE like the rest of this file.
# --------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train.py:327-329): total L2 norm over all gradients; scale by max_norm / (norm + 1e-6)
    when that is < 1.  Returns (total_norm, clipped gradients)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def radam_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One radam.py step (radam.py:52-120) on every tensor of `params` (in place); `state` = {"step": int, "m": {k: tensor},
    "v": {k: tensor}} (exp_avg / exp_avg_sq, radam.py:63-66), created empty by the caller."""
    import math
    beta1, beta2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    beta2_t = beta2 ** t
    n_sma_max = 2 / (1 - beta2) - 1                                             # radam.py:86
    n_sma = n_sma_max - 2 * t * beta2_t / (1 - beta2_t)                         # radam.py:87-89
    if n_sma >= 5:                                                              # radam.py:93-101
        step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) \
            / (1 - beta1 ** t)
    else:                                                                       # radam.py:102-103
        step_size = lr / (1 - beta1 ** t)
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m", {}).setdefault(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).setdefault(k, torch.zeros_like(p))
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)                           # radam.py:78
        m.mul_(beta1).add_(g, alpha=1 - beta1)                                  # radam.py:79
        if weight_decay != 0:
            p.add_(p, alpha=-weight_decay * lr)                                 # radam.py:106-109
        if n_sma >= 5:
            p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)                 # radam.py:112-114
        else:
            p.add_(m, alpha=-step_size)                                         # radam.py:116
