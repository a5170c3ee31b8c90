"""Host-side mirror of the reference model interface (flowtron.py): same class names, ctor
arguments, submodule attribute names (hence identical state_dict keys) and
forward/infer signatures -- every tensor op underneath is a HIP kernel reached through the
C ABI (flowtron_amd.ops).  torch.nn modules are used only as PARAMETER CONTAINERS (so that
checkpoints keep their layout); their torch forward() is never called.

Reference lines are cited per class.  There is no CPU path: calling forward/infer with
CPU tensors raises (the CPU oracle is oracle/flowtron_oracle.py, test infrastructure).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
from torch import nn

from . import _lib as L
from . import ops

# NOTE: train.py-format checkpoints pickle the whole nn.Module; torch >= 2.6 refuses them under the default
# weights_only=True.  Running the reference's unmodified train.py resume path (train.py:87,112) therefore needs
# TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 in the environment of THAT process -- this package does not change
# the process-wide torch.load safety default at import.


# --------------------------------------------------------------------------
# thin wrappers (names are baked into state_dict keys) -- flowtron.py:278-309, 95-126, 453-464
# --------------------------------------------------------------------------
class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain="linear"):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x, act=L.ACT_NONE, rowmap=None, fill="y+dx"):
        return ops.linear(x, self.linear_layer.weight, self.linear_layer.bias, act, rowmap=rowmap, fill=fill)


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1,
                 bias=True, w_init_gain="linear"):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        assert stride == 1 and dilation == 1, "the hot path only uses stride-1, undilated convolutions"
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


class MaskedInstanceNorm1d(nn.Module):
    """Parameter holder for the length-masked instance norm (flowtron.py:95-126):
    affine, no running statistics -> state_dict has exactly `weight` and `bias`."""

    def __init__(self, num_features, eps=1e-5, affine=True, **_):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))


class DenseLayer(nn.Module):
    def __init__(self, in_dim=1024, sizes=(1024, 1024)):
        super().__init__()
        sizes = list(sizes)
        in_sizes = [in_dim] + sizes[:-1]
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=True) for i, o in zip(in_sizes, sizes)])

    def forward(self, x, rowmap=None):
        for lin in self.layers:
            # GEMM with bias + tanh fused in the epilogue; with a row map only valid frames are multiplied, and since the only
            # reader of a dense output is the next compact GEMM, padded rows stay unwritten (fill = "")
            x = lin(x, L.ACT_TANH, rowmap=rowmap, fill="")
        return x


# --------------------------------------------------------------------------
# Encoder -- flowtron.py:467-525
# --------------------------------------------------------------------------
class Encoder(nn.Module):
    def __init__(self, encoder_n_convolutions=3, encoder_embedding_dim=512, encoder_kernel_size=5,
                 norm_fn=MaskedInstanceNorm1d):
        super().__init__()
        convs = []
        for _ in range(encoder_n_convolutions):
            convs.append(nn.Sequential(
                ConvNorm(encoder_embedding_dim, encoder_embedding_dim, kernel_size=encoder_kernel_size, stride=1,
                         padding=int((encoder_kernel_size - 1) / 2), dilation=1, w_init_gain="relu"),
                norm_fn(encoder_embedding_dim, affine=True)))
        self.convolutions = nn.ModuleList(convs)
        self.lstm = nn.LSTM(encoder_embedding_dim, int(encoder_embedding_dim / 2), 1, batch_first=True,
                            bidirectional=True)
        self.dropout_masks = None     # test hook: list of 3 keep-masks [L,B,C] already scaled by 1/(1-p)

    def _run(self, x, lens):
        """x [L,B,C] time-major, lens int32 [B] -> [L,B,C]."""
        # FLOWTRON_ENCODER_F32 = conv | all: the encoder's convolutions (and its BiLSTM) with fp32 operands inside a 16-bit step -- 2 % of
        # the step's FLOPs; measures how much of the 16-bit gradient noise of the embedding / encoder parameters is made HERE
        enc32 = os.environ.get("FLOWTRON_ENCODER_F32", "")
        conv_mode = L.FT_F32 if enc32 in ("conv", "all") else None
        lstm_mode = L.FT_F32 if enc32 == "all" else None
        m16 = L.mfma_mode()
        if L.is16(m16) and enc32 in ("fwd", "dx", "dw", "fwd+dx"):      # one GEMM of the three in fp32 operands (which one makes the noise?)
            conv_mode = (L.FT_F32 if "fwd" in enc32 else m16, L.FT_F32 if "dx" in enc32 else m16, L.FT_F32 if "dw" in enc32 else m16)
        elif L.is16(m16) and enc32 in ("", "split"):
            # DEFAULT in the 16-bit modes (round 5): the convolutions' FORWARD products at fp32 grade from split images (x = hi + lo:
            # one 16-bit GEMM over [hi | lo | hi] x [hi | hi | lo], ops.Bf16Image.split3), backward in the step's 16-bit format.
            # Measured on the bench batch: the 16-bit rounding of these three forward GEMMs -- renormalised by the instance norm
            # behind each -- was what made the text-embedding / encoder gradients deviate 0.11 from the fp32 oracle (the real
            # reference under bf16 autocast: 0.17); with fp32-grade forward products the worst gradient of the model reads 0.008.
            # fp32 operands for the input or the weight gradients instead change nothing (0.113).  FLOWTRON_ENCODER_F32=off: 16-bit.
            conv_mode = ("split3", m16, m16)
        drawn = None
        if self.dropout_masks is None and self.training:
            # F.dropout(p=0.5) keep-masks (flowtron.py:502) of ALL conv layers in one draw: two kernels per step instead of four per layer
            # (every layer keeps the [L,B,C] shape: the convolutions are C -> C)
            drawn = torch.empty((len(self.convolutions),) + tuple(x.shape), device=x.device, dtype=x.dtype).bernoulli_(0.5).mul_(2.0)
        for i, (conv, norm) in enumerate(self.convolutions):
            keep = None
            if self.dropout_masks is not None:
                keep = self.dropout_masks[i]
            elif drawn is not None:
                keep = drawn[i]
            x = ops.conv_norm_relu(x, lens, conv.conv.weight, conv.conv.bias, norm.weight, norm.bias, keep, norm.eps, mode=conv_mode)
        p = self.lstm
        # both directions in one launch chain when the fragment path applies (ops.bilstm_layer)
        return ops.bilstm_layer(x, lens, (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0),
                                (p.weight_ih_l0_reverse, p.weight_hh_l0_reverse, p.bias_ih_l0_reverse, p.bias_hh_l0_reverse), mode=lstm_mode)

    def forward(self, x, in_lens):
        """x [B,C,L] (reference layout) -> [B,L,C]"""
        xt = x.permute(2, 0, 1).contiguous()
        return self._run(xt, ops.lens32(in_lens)).transpose(0, 1)

    def infer(self, x):
        xt = x.permute(2, 0, 1).contiguous()
        lens = torch.full((xt.shape[1],), xt.shape[0], dtype=torch.int32, device=xt.device)
        return self._run(xt, lens).transpose(0, 1)


# --------------------------------------------------------------------------
# Attention -- flowtron.py:528-592
# --------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, n_mel_channels=80, n_speaker_dim=128, n_text_channels=512, n_att_channels=128, temperature=1.0):
        super().__init__()
        self.temperature = temperature
        self.query = LinearNorm(n_mel_channels, n_att_channels, bias=False, w_init_gain="tanh")
        self.key = LinearNorm(n_text_channels + n_speaker_dim, n_att_channels, bias=False, w_init_gain="tanh")
        self.value = LinearNorm(n_text_channels + n_speaker_dim, n_att_channels, bias=False, w_init_gain="tanh")
        self.v = LinearNorm(n_att_channels, 1, bias=False, w_init_gain="tanh")
        self.score_mask_value = -float("inf")

    def forward(self, queries, keys, values, in_lens32, attn_prior=None, rowmap=None):
        """queries [T,B,H], keys/values source [L,B,E] -> ctx [T,B,A], attn [B,T,L], attn_logprob [B,T,L]."""
        mode = L.mfma_mode()
        K = ops.linear(keys, self.key.linear_layer.weight, None, mode=mode)
        V = ops.linear(values, self.value.linear_layer.weight, None, mode=mode)
        # the score kernel walks every frame (the reference returns the attention of padded frames too): padded query rows are
        # filled with the utterance's first padded row (fill "y"); the query's input gradient only feeds guarded consumers
        Q = ops.linear(queries, self.query.linear_layer.weight, None, mode=mode, rowmap=rowmap, fill="y")
        attn, logprob = ops.AttentionScoresFn.apply(Q, K, self.v.linear_layer.weight, in_lens32, attn_prior,
                                                    self.temperature)
        ctx = ops.ContextFn.apply(attn, V, mode)
        return ctx, attn, logprob


# --------------------------------------------------------------------------
# AttentionConditioningLayer -- flowtron.py:129-152 (location-sensitive / cumulative attention features)
# --------------------------------------------------------------------------
class AttentionConditioningLayer(nn.Module):
    """Conv1d(2->32,k5)+ReLU+Conv1d(32->attention_dim,k3)+Sigmoid over [cumulative attention ; previous attention].
    Same registration as the reference (the nn.Sequential re-registers both convs, so the state_dict carries the
    `location_conv_*` AND `conv_layers.{0,2}` aliases).  Evaluated as im2col + GEMM with the activation fused in the
    GEMM epilogue, directly in the time-major [L,B,C] layout the key modulation needs."""

    def __init__(self, input_dim=2, attention_n_filters=32, attention_kernel_sizes=(5, 3), attention_dim=640):
        super().__init__()
        self.location_conv_hidden = ConvNorm(input_dim, attention_n_filters, kernel_size=attention_kernel_sizes[0],
                                             padding=None, bias=True, stride=1, dilation=1, w_init_gain="relu")
        self.location_conv_out = ConvNorm(attention_n_filters, attention_dim, kernel_size=attention_kernel_sizes[1],
                                          padding=None, bias=True, stride=1, dilation=1, w_init_gain="sigmoid")
        self.conv_layers = nn.Sequential(self.location_conv_hidden, nn.ReLU(), self.location_conv_out, nn.Sigmoid())

    def forward(self, attn_cat_lbc, full_lens):
        """attn_cat_lbc [L,B,2] time-major, full_lens int32 [B] (== L: Conv1d zero-pads only at the ends) -> [L,B,attention_dim]."""
        c1, c2 = self.location_conv_hidden.conv, self.location_conv_out.conv
        col = ops.Im2colFn.apply(attn_cat_lbc, full_lens, c1.weight.shape[2])
        h = ops.linear(col, c1.weight.reshape(c1.weight.shape[0], -1), c1.bias, act=L.ACT_RELU)
        col = ops.Im2colFn.apply(h, full_lens, c2.weight.shape[2])
        return ops.linear(col, c2.weight.reshape(c2.weight.shape[0], -1), c2.bias, act=L.ACT_SIGMOID)


# --------------------------------------------------------------------------
# AR_Step / AR_Back_Step -- flowtron.py:645-828, 595-642
# --------------------------------------------------------------------------
class AR_Step(nn.Module):
    def __init__(self, n_mel_channels, n_speaker_dim, n_text_channels, n_in_channels, n_hidden, n_attn_channels,
                 n_lstm_layers, add_gate, use_cumm_attention):
        super().__init__()
        if n_lstm_layers < 1:
            raise ValueError("n_lstm_layers must be >= 1")
        self.n_lstm_layers = int(n_lstm_layers)       # any depth: training = one recurrence + one batched projection per layer; decode:
        self.use_cumm_attention = use_cumm_attention  # depth 2 on the persistent launch, any other depth on the staged chain
        self.conv = nn.Conv1d(n_hidden, 2 * n_mel_channels, 1)
        self.conv.weight.data = 0.0 * self.conv.weight.data
        self.conv.bias.data = 0.0 * self.conv.bias.data
        self.lstm = nn.LSTM(n_hidden + n_attn_channels, n_hidden, n_lstm_layers)
        self.attention_lstm = nn.LSTM(n_mel_channels, n_hidden)
        self.attention_layer = Attention(n_hidden, n_speaker_dim, n_text_channels, n_attn_channels)
        if self.use_cumm_attention:
            self.attn_cond_layer = AttentionConditioningLayer(input_dim=2, attention_n_filters=32,
                                                              attention_kernel_sizes=[5, 3],
                                                              attention_dim=n_text_channels + n_speaker_dim)
        self.dense_layer = DenseLayer(in_dim=n_hidden, sizes=[n_hidden, n_hidden])
        if add_gate:
            self.gate_threshold = 0.5
            self.gate_layer = LinearNorm(n_hidden + n_attn_channels, 1, bias=True, w_init_gain="sigmoid")
        self._decode_work = None

    def __getstate__(self):
        """train.py pickles the whole module into its checkpoints (train.py:131-141): the decode scratch (buffers per utterance
        shape, the 54 MB weight image, the hand-off granules) is runtime state, not model state."""
        d = self.__dict__.copy()
        for k in ("_decode_bufs", "_decode_wimg", "_decode_gran"):
            d.pop(k, None)
        d["_decode_work"] = None
        return d

    def forward(self, mel, text, in_lens32, out_lens32, attn_prior=None, rowmap=None):
        """Teacher-forced flow. mel [T,B,M], text = encoder outputs [L,B,E].
        Returns (z [T,B,M], log_s [T,B,M], gates [T,B,1] | None, attn [B,T,L], attn_logprob [B,T,L]).
        rowmap (ops.RowMap over out_lens32): the batched GEMMs multiply valid frames only (pack-by-length, flowtron.py:689-694)."""
        T, B, M = mel.shape
        mode = L.mfma_mode()
        rm = rowmap
        mel0 = torch.cat([mel.new_zeros(1, B, M), mel[:-1]], 0)              # flowtron.py:726-729
        a = self.attention_lstm
        # fill "dx": d(mel0) flows into the PREVIOUS flow's z gradient, whose separator rows (first padded frame of an utterance)
        # the compact weight-gradient GEMMs of that flow do read -- padded rows must be zeros, not unwritten memory (T*B*80 floats)
        h_att = ops.lstm_layer(mel0, out_lens32, a.weight_ih_l0, a.weight_hh_l0, a.bias_ih_l0, a.bias_hh_l0, mode=mode, rowmap=rm,
                               fill="dx")
        if self.use_cumm_attention:
            ctx, attn, logprob = self.run_cumm_attn_sequence(h_att, text, in_lens32)   # drops the prior like flowtron.py:742-743
        else:
            ctx, attn, logprob = self.attention_layer(h_att, text, text, in_lens32, attn_prior, rowmap=rm)
        gates = None
        p = self.lstm
        # the gate layer reads [h_att ; ctx] like the decoder LSTM's input projection: where that projection runs over a
        # concatenated operand image, the N = 1 gate projection is taken from the same image (ops.LinearGateFn)
        g = self.gate_layer.linear_layer if hasattr(self, "gate_layer") else None
        persist = (ops.lstm_persist_groups(B, p.weight_hh_l0.shape[1], False, mode, mel.device)
                   or bool(ops.lstm_persist_slices(B, p.weight_hh_l0.shape[1], False, mode, mel.device))     # (B > 32: sliced launches)
                   or ops.lstm_pad_width(B, p.weight_hh_l0.shape[1], False, mode, mel.device, T))             # (H < 1024: zero-padded twin)
        fuse_gate = g is not None and (self.n_lstm_layers != 2 or persist or not ops.lstm2_supported(B, p.weight_hh_l0.shape[1], mode))
        if g is not None and not fuse_gate:
            gates = ops.linear([h_att, ctx], g.weight, g.bias, mode=mode)     # Linear over [h_att ; ctx], no concat

        def first_layer():
            r = ops.lstm_layer(h_att, out_lens32, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0, mode=mode,
                               xs_extra=[ctx], rowmap=rm, fill="dx", gate=(g.weight, g.bias) if fuse_gate else None)
            return r if fuse_gate else (r, gates)
        if self.n_lstm_layers != 2:
            # any other depth (the config schema splats n_lstm_layers into nn.LSTM, flowtron.py:655): the same per-layer pair --
            # batched input projection + one recurrence (persistent where its geometry applies) -- layer after layer
            h, gates = first_layer()
            for l in range(1, self.n_lstm_layers):
                h = ops.lstm_layer(h, out_lens32, getattr(p, "weight_ih_l%d" % l), getattr(p, "weight_hh_l%d" % l),
                                   getattr(p, "bias_ih_l%d" % l), getattr(p, "bias_hh_l%d" % l), mode=mode, rowmap=rm)
        elif persist and rm is not None and ops.decoder_pair_chunks(B, p.weight_hh_l0.shape[1], mode, mel.device, T):
            # both layers CONCURRENTLY on four XCDs each, layer 1 one time chunk behind layer 0, the chunk's input projection between
            # the launches (ops.DecoderPairFn, csrc/lstm_roles.hip): the pair's chain of 2 T dependent steps becomes (1 + 1/n) T
            h, gates_f = ops.decoder_pair(h_att, out_lens32, p, mode, [ctx], rm, "dx", (g.weight, g.bias) if fuse_gate else None,
                                          ops.decoder_pair_chunks(B, p.weight_hh_l0.shape[1], mode, mel.device, T))
            if fuse_gate:
                gates = gates_f
        elif persist:
            # two persistent single-layer recurrences (csrc/lstm_persist.hip, ~2 us per step each) with layer 1's input
            # projection as one batched GEMM between them: faster than the two-layer wavefront launch chain (~8 us per step)
            # (the context gradient feeds the attention backward, which reduces over ALL frames: zero its padded rows)
            h, gates = first_layer()
            h = ops.lstm_layer(h, out_lens32, p.weight_ih_l1, p.weight_hh_l1, p.bias_ih_l1, p.bias_hh_l1, mode=mode, rowmap=rm)
        elif ops.lstm2_supported(B, p.weight_hh_l0.shape[1], mode):
            # both decoder layers as one software-wavefront launch chain (csrc/lstm2.hip)
            gx0 = ops.LinearFn.apply(p.weight_ih_l0, p.bias_ih_l0 + p.bias_hh_l0, L.ACT_NONE, mode, None, "", h_att, ctx)
            h = ops.LSTM2SeqFn.apply(gx0, p.weight_hh_l0, p.weight_ih_l1, p.bias_ih_l1, p.bias_hh_l1, p.weight_hh_l1, out_lens32, mode)
        else:
            h, gates = first_layer()
            h = ops.lstm_layer(h, out_lens32, p.weight_ih_l1, p.weight_hh_l1, p.bias_ih_l1, p.bias_hh_l1, mode=mode, rowmap=rm)
        h = self.dense_layer(h, rowmap=rm)
        # the coupling output is returned for every frame (z, log_s of padded frames are the reference's defined junk): fill "y"
        out = ops.linear(h, self.conv.weight.reshape(self.conv.weight.shape[0], -1), self.conv.bias, mode=mode, rowmap=rm, fill="y")
        z = ops.AffineFn.apply(out, mel)                                      # z = exp(log_s) * mel + b
        log_s = out[..., :M]
        return z, log_s, gates, attn, logprob

    def run_cumm_attn_sequence(self, h_att, text, in_lens32):
        """flowtron.py:697-723: the location features of frame i depend on the attention of frames < i, so the frames are walked
        in order -- by the LIBRARY (ops.CummAttnSeqFn -> csrc/cumm_attn.hip: one C-ABI call per flow forward, one backward), not by
        Python: location convolutions as im2col + GEMM, key modulation, the per-frame key projection, a fused score / softmax /
        context / running-sum kernel.  FLOWTRON_CUMM_LOOP=python keeps the per-frame autograd walk (every operation a HIP kernel
        too; ~250 us of host time per frame) as the yardstick the tests compare the fused path with."""
        mode = L.mfma_mode()
        att = self.attention_layer
        V = ops.linear(text, att.value.linear_layer.weight, None, mode=mode)
        Q = ops.linear(h_att, att.query.linear_layer.weight, None, mode=mode)
        if os.environ.get("FLOWTRON_CUMM_LOOP", "fused") != "python":
            c = self.attn_cond_layer
            return ops.CummAttnSeqFn.apply(Q, V, text, att.key.linear_layer.weight, att.v.linear_layer.weight,
                                           c.location_conv_hidden.conv.weight, c.location_conv_hidden.conv.bias,
                                           c.location_conv_out.conv.weight, c.location_conv_out.conv.bias,
                                           in_lens32, float(att.temperature), mode)
        T, B, _ = h_att.shape
        Lk = text.shape[0]
        full = torch.full((B,), Lk, dtype=torch.int32, device=text.device)
        cumm = text.new_zeros(Lk, B)
        prev = text.new_zeros(Lk, B)
        ctxs, attns, lps = [], [], []
        for i in range(T):
            cond = self.attn_cond_layer(torch.stack([cumm, prev], 2), full)                 # [L,B,E]
            K = ops.linear(ops.MulFn.apply(text, cond), att.key.linear_layer.weight, None, mode=mode)
            a, lp = ops.AttentionScoresFn.apply(Q[i:i + 1], K, att.v.linear_layer.weight, in_lens32, None, att.temperature)
            ctxs.append(ops.ContextFn.apply(a, V, mode))
            attns.append(a)
            lps.append(lp)
            prev = a[:, 0, :].t()
            cumm = ops.AddFn.apply(cumm, prev)
        return torch.cat(ctxs, 0), torch.cat(attns, 1), torch.cat(lps, 1)

    def infer(self, residual, text, attns=None, attn_prior=None, use_graph=None):
        """Sequential inverse (flowtron.py:775-828). residual [N,B,M], text [L,B,E].
        Returns (mel [N',B,M], list of N' attention rows [B,1,L]).
        The decode kernels are batch-1 chains of GEMVs (inference.py decodes one utterance); B > 1 (round 5: the reference's loop takes
        any batch, flowtron.py:775-828) decodes the utterances one after the other through the same kernels -- AR decode does not
        shard, so a batch is B replicas.  With a gate layer every utterance stops at its OWN frame (the reference's
        `if sigmoid(gate) > thr` raises for B > 1); frames behind an utterance's stop are zero, N' = the longest."""
        N, B, M = residual.shape
        if B != 1:
            outs = []
            for b_ in range(B):
                pr = None if attn_prior is None else attn_prior[b_:b_ + 1]
                if attns is not None:
                    raise ValueError("forced alignments (attns=) are taken one utterance at a time")
                outs.append(self.infer(residual[:, b_:b_ + 1].contiguous(), text[:, b_:b_ + 1].contiguous(), None, pr, use_graph))
            n = max(o[0].shape[0] for o in outs)
            mel = residual.new_zeros(n, B, M)
            Lk = text.shape[0]
            att = residual.new_zeros(n, B, 1, Lk)
            for b_, (m_, rows) in enumerate(outs):
                mel[:m_.shape[0], b_] = m_[:, 0]
                if rows:
                    att[:len(rows), b_] = torch.stack([r_.reshape(1, Lk) for r_ in rows])
            return mel, [att[t_] for t_ in range(n)]
        L.require_cuda(residual, text)
        Lk = text.shape[0]
        att = self.attention_layer
        H = self.lstm.weight_hh_l0.shape[1]
        A = att.key.linear_layer.weight.shape[0]
        dev = residual.device
        # persistent per-(N, L) buffers: the decode hipGraph bakes every pointer, so stable addresses = one capture,
        # replayed for every later utterance of this shape
        key = (N, Lk, str(dev))
        if not hasattr(self, "_decode_bufs"):
            import collections
            self._decode_bufs = collections.OrderedDict()       # LRU over (N, L, device): a server with varied lengths stays bounded
        bufs = self._decode_bufs.get(key)
        if bufs is not None:
            self._decode_bufs.move_to_end(key)
        else:
            while len(self._decode_bufs) >= 8:
                self._decode_bufs.popitem(last=False)
            f32 = dict(device=dev, dtype=torch.float32)
            bufs = self._decode_bufs[key] = dict(K=torch.empty(Lk, A, **f32), V=torch.empty(Lk, A, **f32), res=torch.empty(N, M, **f32),
                                                 mel=torch.empty(N, M, **f32), attn=torch.empty(N, Lk, **f32),
                                                 n_done=torch.zeros(1, device=dev, dtype=torch.int32))
        K, V, res, mel_out, attn_out, n_done = (bufs[k] for k in ("K", "V", "res", "mel", "attn", "n_done"))
        K.copy_(ops.linear(text, att.key.linear_layer.weight, None).reshape(Lk, A))
        V.copy_(ops.linear(text, att.value.linear_layer.weight, None).reshape(Lk, A))
        res.copy_(residual.reshape(N, M))
        attn_out.zero_()
        prior_rows = forced_rows = None
        if attn_prior is not None:              # [1, N, L] (flowtron.py:799 takes attn_prior[:, i])
            prior_rows = attn_prior.reshape(-1, Lk)[:N].contiguous().float()
            assert prior_rows.shape[0] == N, "attn_prior must cover every residual frame"
        if attns is not None:                   # N rows of [L] (flowtron.py:798 takes attns[i])
            forced_rows = (torch.stack([a_.reshape(-1) for a_ in attns]) if isinstance(attns, (list, tuple)) else attns.reshape(-1, Lk))
            forced_rows = forced_rows[:N].contiguous().float()
            assert forced_rows.shape == (N, Lk), "attns must hold one attention row per residual frame"
        L.require_cuda(prior_rows, forced_rows)
        cumm = self.use_cumm_attention
        E = text.shape[2]
        enc2d = text.reshape(Lk, E).contiguous()
        nl = self.n_lstm_layers
        nbytes = L.lib().ft_decode_workspace_bytes(Lk, H, A, M, E if cumm else 1, nl)
        work = bufs.get("work")
        if work is None or work.numel() < nbytes:
            work = bufs["work"] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        has_gate = hasattr(self, "gate_layer")
        if use_graph is None:
            use_graph = os.environ.get("FLOWTRON_DECODE_GRAPH", "1") != "0"
        a, p, d = self.attention_lstm, self.lstm, self.dense_layer.layers
        keep = [K, V, res, enc2d, prior_rows, forced_rows]   # keep temporaries alive until the launch is enqueued
        cc = self.attn_cond_layer if cumm else None
        args = L.DecodeArgs(
            L.ptr(a.weight_ih_l0), L.ptr(a.weight_hh_l0), L.ptr(a.bias_ih_l0), L.ptr(a.bias_hh_l0),
            L.ptr(att.query.linear_layer.weight), L.ptr(att.v.linear_layer.weight), L.ptr(K), L.ptr(V),
            L.ptr(p.weight_ih_l0), L.ptr(p.weight_hh_l0), L.ptr(p.bias_ih_l0), L.ptr(p.bias_hh_l0),
            *([L.ptr(p.weight_ih_l1), L.ptr(p.weight_hh_l1), L.ptr(p.bias_ih_l1), L.ptr(p.bias_hh_l1)] if nl >= 2 else [None] * 4),
            L.ptr(d[0].linear_layer.weight), L.ptr(d[0].linear_layer.bias),
            L.ptr(d[1].linear_layer.weight), L.ptr(d[1].linear_layer.bias),
            L.ptr(self.conv.weight), L.ptr(self.conv.bias),
            L.ptr(self.gate_layer.linear_layer.weight) if has_gate else None,
            L.ptr(self.gate_layer.linear_layer.bias) if has_gate else None,
            L.ptr(res), L.ptr(mel_out), L.ptr(attn_out), L.ptr(n_done), L.ptr(work),
            work.numel(), N, Lk, H, A, M, float(att.temperature),
            float(self.gate_threshold) if has_gate else 2.0, int(bool(use_graph)),
            L.ptr(cc.location_conv_hidden.conv.weight) if cumm else None, L.ptr(cc.location_conv_hidden.conv.bias) if cumm else None,
            L.ptr(cc.location_conv_out.conv.weight) if cumm else None, L.ptr(cc.location_conv_out.conv.bias) if cumm else None,
            L.ptr(att.key.linear_layer.weight) if cumm else None, L.ptr(enc2d) if cumm else None, E,
            L.ptr(prior_rows), L.ptr(forced_rows), None, 0, None, None, nl, None)
        if nl > 2:
            # decoder layers beyond the second (any n_lstm_layers, flowtron.py:654-655): a host array of their four tensors each
            ptrs = [L.ptr(getattr(p, "%s_l%d" % (nm, k))) for k in range(2, nl) for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            extra = (C.c_void_p * len(ptrs))(*ptrs)
            keep.append(extra)
            args.extra_layers = C.cast(extra, C.c_void_p)
        persist = None
        if L.is16(L.mfma_mode()):               # 16-bit operand modes: bf16 images of the weights (half the bytes; fp32 activations)
            nb = L.lib().ft_decode_wimg_bytes(H, A, M)
            wimg = getattr(self, "_decode_wimg", None)           # one image buffer per flow, whatever the utterance shape
            if wimg is None or wimg.numel() < nb or wimg.device != dev:
                wimg = self._decode_wimg = torch.empty(nb, device=dev, dtype=torch.uint8)
            args.wimg, args.wimg_bytes = L.ptr(wimg), wimg.numel()
        if nl == 2 and os.environ.get("FLOWTRON_DECODE_PERSIST", "1") != "0" and ops.persist_usable(dev):
            # one persistent launch per flow (csrc/decode.hip dec_persist_k) where its geometry applies: 16-bit weight images fully
            # register-resident, or (fp32 mode = the reference's inference precision, round 4) the fp32 originals with the recurrent
            # matrices resident and the rest streamed from the L2 / Infinity Cache
            gran = getattr(self, "_decode_gran", None)
            if gran is None or gran.device != dev:
                gran = self._decode_gran = torch.empty(L.lib().ft_decode_persist_gran_bytes(), device=dev, dtype=torch.uint8)
            persist = ops.persist_status(dev)
            args.persist_gran, args.persist_status = L.ptr(gran), L.ptr(persist)
        L.check(L.lib().ft_decode_flow(C.byref(args), L.stream()), "ft_decode_flow")
        n = int(n_done.item()) if has_gate else N          # single host read per flow (the reference syncs every frame)
        if persist is not None and not ops.check_persist_status(raise_on_failure=False):
            # the one-launch decode did not complete (grid not co-resident): logged once by ops, the device is switched to the
            # launch-per-step / staged kernels, and THIS flow is decoded again on the staged chain -- the caller never sees it
            args.persist_gran, args.persist_status = None, None
            attn_out.zero_()
            L.check(L.lib().ft_decode_flow(C.byref(args), L.stream()), "ft_decode_flow")
            n = int(n_done.item()) if has_gate else N
        del keep
        mel = mel_out[:n].clone().reshape(n, 1, M)          # the persistent buffers are overwritten by the next call
        attn_all = attn_out[:n].clone()
        attn_rows = [attn_all[i].reshape(1, 1, Lk) for i in range(n)]
        return mel, attn_rows


class AR_Back_Step(nn.Module):
    def __init__(self, n_mel_channels, n_speaker_dim, n_text_dim, n_in_channels, n_hidden, n_attn_channels,
                 n_lstm_layers, add_gate, use_cumm_attention):
        super().__init__()
        self.ar_step = AR_Step(n_mel_channels, n_speaker_dim, n_text_dim, n_mel_channels + n_speaker_dim, n_hidden,
                               n_attn_channels, n_lstm_layers, add_gate, use_cumm_attention)
        self.ar_step._time_reversed = True

    def forward(self, mel, text, in_lens32, out_lens32, attn_prior=None, rowmap=None):
        # flip + per-sample roll (flowtron.py:606-613) == the reverse-by-length involution, one gather kernel, no host syncs
        mel = ops.reverse_by_length(mel, out_lens32, True)
        if attn_prior is not None:
            attn_prior = ops.reverse_by_length(attn_prior, out_lens32, False)
        z, log_s, gates, attn, logprob = self.ar_step(mel, text, in_lens32, out_lens32, attn_prior, rowmap=rowmap)
        z = ops.reverse_by_length(z, out_lens32, True)
        return z, log_s, gates, attn, logprob

    def infer(self, residual, text, attns=None, attn_prior=None):
        if attn_prior is not None:
            attn_prior = torch.flip(attn_prior, (1,))                     # flowtron.py:631-633 (batch 1: flip only)
        out, attn = self.ar_step.infer(torch.flip(residual, (0,)), text, attns, attn_prior=attn_prior)
        return torch.flip(out, (0,)), attn


# --------------------------------------------------------------------------
# Flowtron -- flowtron.py:831-961
# --------------------------------------------------------------------------
class Flowtron(nn.Module):
    def __init__(self, n_speakers, n_speaker_dim, n_text, n_text_dim, n_flows, n_mel_channels, n_hidden,
                 n_attn_channels, n_lstm_layers, use_gate_layer, mel_encoder_n_hidden, n_components,
                 fixed_gaussian, mean_scale, dummy_speaker_embedding, use_cumm_attention):
        super().__init__()
        norm_fn = MaskedInstanceNorm1d
        self.speaker_embedding = nn.Embedding(n_speakers, n_speaker_dim)
        self.embedding = nn.Embedding(n_text, n_text_dim)
        self.flows = nn.ModuleList()
        self.encoder = Encoder(norm_fn=norm_fn, encoder_embedding_dim=n_text_dim)
        self.dummy_speaker_embedding = dummy_speaker_embedding
        if n_components > 1:
            raise NotImplementedError("the Gaussian-mixture prior (n_components > 1) is outside the hot path "
                                      "(config.json:60 uses 0); see DESIGN.md")
        for i in range(n_flows):
            add_gate = bool(i == (n_flows - 1) and use_gate_layer)
            cls = AR_Step if i % 2 == 0 else AR_Back_Step
            self.flows.append(cls(n_mel_channels, n_speaker_dim, n_text_dim, n_mel_channels + n_speaker_dim,
                                  n_hidden, n_attn_channels, n_lstm_layers, add_gate, use_cumm_attention))

    def _encode(self, speaker_ids, text, in_lens):
        """embeddings + encoder + speaker concat -> [L,B,E] (flowtron.py:872-887)."""
        L.require_cuda(speaker_ids, text, self.embedding.weight)
        if self.dummy_speaker_embedding:
            speaker_ids = speaker_ids * 0
        B, Lt = text.shape
        # speaker row broadcast over L by the gather kernel itself (ids repeated), so its backward is the
        # same atomic scatter-add kernel (flowtron.py:886-887 expand + cat)
        spk_ids = speaker_ids.reshape(1, -1).expand(Lt, -1).reshape(-1)
        spk = ops.embedding(spk_ids, self.speaker_embedding.weight, run_stride=B).reshape(Lt, B, -1)   # [L,B,S]; backward: one atomic per 32 positions
        emb = ops.embedding(text.t().contiguous(), self.embedding.weight).reshape(Lt, B, -1)   # time-major [L,B,C]
        if in_lens is None:
            lens = torch.full((B,), Lt, dtype=torch.int32, device=text.device)
        else:
            lens = ops.lens32(in_lens)
        enc = self.encoder._run(emb, lens)
        return torch.cat([enc, spk], 2), lens

    def forward(self, mel, speaker_ids, text, in_lens, out_lens, attn_prior=None):
        """mel [B,M,T], speaker_ids [B], text [B,L] (sorted by in_lens desc), in_lens/out_lens [B],
        attn_prior [B,T,L] | None -> the reference's 8-tuple (flowtron.py:898-899)."""
        L.require_cuda(mel, text, in_lens, out_lens, attn_prior)
        ops.weight_images_begin(self)                # (the weights this model's previous forward rounded: one launch at the first request)
        ops.bias_sums_begin(self._lstm_bias_pairs())  # (b_ih + b_hh of every LSTM: one multi-tensor add)
        try:
            return self._forward(mel, speaker_ids, text, in_lens, out_lens, attn_prior)
        finally:
            ops.bias_sums_end()
            ops.weight_images_end(self)

    def _lstm_bias_pairs(self):
        """[(b_ih, b_hh)] groups, one per flow / for the encoder: a group is one autograd node, and its parameters' gradients become
        ready together -- per flow, so that the per-flow gradient buckets of the DP overlap regime still complete in backward order"""
        groups = {}
        for mname, m in self.named_modules():
            if isinstance(m, torch.nn.LSTM):
                key = ".".join(mname.split(".")[:2]) if mname.startswith("flows.") else mname.split(".")[0]
                for name, p in m.named_parameters(recurse=False):
                    if name.startswith("bias_ih"):
                        groups.setdefault(key, []).append((p, getattr(m, "bias_hh" + name[len("bias_ih"):])))
        return list(groups.values())

    def _forward(self, mel, speaker_ids, text, in_lens, out_lens, attn_prior=None):
        enc, in32 = self._encode(speaker_ids, text, in_lens)
        out32 = ops.lens32(out_lens)
        x = mel.permute(2, 0, 1).contiguous().float()
        if attn_prior is not None:
            attn_prior = attn_prior.float()
        # pack-by-length for the batched GEMMs of the 16-bit operand modes (one row map per forward, built on the device)
        rm = ops.row_map(out32, x.shape[0], x.shape[1]) if L.is16(L.mfma_mode()) else None

        def run_flows(x, enc):
            log_s_list, attns_list, attns_logprob_list = [], [], []
            gate = None
            for flow in self.flows:
                x, log_s, gate, attn, logprob = flow(x, enc, in32, out32, attn_prior, rowmap=rm)
                log_s_list.append(log_s)
                attns_list.append(attn)
                attns_logprob_list.append(logprob)
            return x, log_s_list, gate, attns_list, attns_logprob_list

        x0 = x
        x, log_s_list, gate, attns_list, attns_logprob_list = run_flows(x0, enc)
        if not torch.is_grad_enabled() and ops.PERSIST_LAUNCHES:
            # forward-only pass (validation, train.py:143-202): no optimizer step will look at the persistent kernels' status word,
            # so a launch that timed out would hand back garbage silently.  One host read (the validation loop reads its losses
            # with .item() anyway); on a failure -- this pass's, or a stale word of an earlier, already dropped training step --
            # the device has been switched to the launch-per-step kernels (warned once) and the flows run AGAIN on those, like
            # AR_Step.infer does: a validation pass must not kill a run that a training step would survive, and under DP the other
            # ranks must not be left waiting at their next collective
            if not ops.check_persist_status(raise_on_failure=False):
                enc, in32 = self._encode(speaker_ids, text, in_lens)          # (the encoder's persistent BiLSTM may be the one that failed)
                x, log_s_list, gate, attns_list, attns_logprob_list = run_flows(x0, enc)
        return x, log_s_list, gate, attns_list, attns_logprob_list, None, None, None

    def infer(self, residual, speaker_ids, text, temperature=1.0, gate_threshold=0.5, attns=None, attn_prior=None):
        """residual [B,M,N], speaker_ids [B] or [B,1], text [B,L] -> (mel [B,M,N'], attention_weights: per flow a list of N' rows
        [B,1,L]).  The decode kernels are batch-1 chains (inference.py decodes one utterance).  A batch (the reference's loop takes any,
        flowtron.py:775-828, 901-930) is decoded utterance by utterance through ALL flows and padded only at the very end: with a gate
        layer every utterance stops at its OWN frame of the last flow (decoded first), and the flows behind it must see exactly that
        utterance's frames -- padding the batch after each flow would put the shorter utterances' pad frames IN FRONT of their real
        ones once the next (reversed) flow flips the tensor.  Frames behind an utterance's stop are zero; N' = the longest."""
        L.require_cuda(residual, text)
        B = residual.shape[0]
        if B > 1:
            if attns is not None:
                raise ValueError("forced alignments (attns=) are taken one utterance at a time")
            sid = speaker_ids.reshape(B, -1)
            outs = [self.infer(residual[b:b + 1], sid[b], text[b:b + 1], temperature, gate_threshold, None,
                               None if attn_prior is None else attn_prior[b:b + 1]) for b in range(B)]
            n = max(int(m.shape[2]) for m, _ in outs)
            mel = residual.new_zeros(B, outs[0][0].shape[1], n, dtype=torch.float32)
            Lk = text.shape[1]
            atts = [residual.new_zeros(n, B, 1, Lk, dtype=torch.float32) for _ in self.flows]
            for b, (m, aws) in enumerate(outs):
                mel[b, :, :m.shape[2]] = m[0]
                for f, rows in enumerate(aws):
                    if rows:
                        atts[f][:len(rows), b] = torch.stack([r.reshape(1, Lk) for r in rows])
            return mel, [[a[t] for t in range(n)] for a in atts]
        with torch.no_grad():
            enc, _ = self._encode(speaker_ids, text, None)
            x = residual.permute(2, 0, 1).contiguous().float()
            attention_weights = []
            for i, flow in enumerate(reversed(self.flows)):
                self.set_temperature_and_gate(flow, temperature, gate_threshold)
                # attns: one forced alignment per flow, in flows order (the reference indexes `reversed(attns)[i]`, which
                # raises TypeError -- flowtron.py:925; the evident intent is implemented)
                x, aw = flow.infer(x, enc, None if attns is None else attns[len(self.flows) - 1 - i], attn_prior=attn_prior)
                attention_weights.append(aw)
            return x.permute(1, 2, 0), attention_weights

    @staticmethod
    def set_temperature_and_gate(flow, temperature, gate_threshold):
        flow = flow.ar_step if hasattr(flow, "ar_step") else flow
        flow.attention_layer.temperature = temperature
        if hasattr(flow, "gate_layer"):
            flow.gate_threshold = gate_threshold


# --------------------------------------------------------------------------
# Loss -- flowtron.py:155-275
# --------------------------------------------------------------------------
class AttentionCTCLoss(nn.Module):
    """flowtron.py:155-182: the reference loops over samples (slice, log_softmax, CTCLoss with target 1..K,
    reduction='mean' per sample, F*B host synchronisations per step).  Here ONE pair of HIP kernels handles the whole
    batch (ft_attn_ctc_fwd/bwd: log-softmax normaliser + banded alpha/beta recursion, one workgroup per sample).
    Device tensors only -- the torch restatement used to pin the batching lives in tests/test_host_cpu.py."""

    def __init__(self, blank_logprob=-1):
        super().__init__()
        self.blank_logprob = blank_logprob

    def forward(self, attn_logprob, in_lens, out_lens):
        """attn_logprob [B,T,L] in natural time order."""
        L.require_cuda(attn_logprob, in_lens, out_lens)
        return ops.AttnCTCFn.apply(attn_logprob, ops.lens32(in_lens), ops.lens32(out_lens), self.blank_logprob)


class FlowtronLoss(nn.Module):
    def __init__(self, sigma=1.0, gm_loss=False, gate_loss=True, use_ctc_loss=False, ctc_loss_weight=0.0,
                 blank_logprob=-1):
        super().__init__()
        if gm_loss:
            raise NotImplementedError("gm_loss (Gaussian-mixture prior) is outside the hot path")
        self.sigma = sigma
        self.gm_loss = gm_loss
        self.gate_loss = gate_loss
        self.use_ctc_loss = use_ctc_loss
        self.ctc_loss_weight = ctc_loss_weight
        self.blank_logprob = blank_logprob
        self.attention_loss = AttentionCTCLoss(blank_logprob=self.blank_logprob)

    def forward(self, model_output, gate_target, in_lengths, out_lengths, is_validation=False):
        z, log_s_list, gate_pred, attn_list, attn_logprob_list = model_output[:5]
        out32 = ops.lens32(out_lengths)
        use_gate = self.gate_loss > 0
        lps = list(attn_logprob_list) if self.use_ctc_loss else []
        if ops.FUSED_LOSS and len(log_s_list) <= 8 and len(lps) <= 8 and all(lp is not None for lp in lps):
            # one autograd node, ten launches forward + backward (ops.FlowtronLossFn); the per-term path below is its restatement
            loss, gate_loss, loss_ctc = ops.FlowtronLossFn.apply(z, gate_pred if use_gate else None, gate_target if use_gate else None,
                                                                 out32, ops.lens32(in_lengths), float(self.sigma),
                                                                 float(self.blank_logprob), len(log_s_list), *log_s_list, *lps)
            if not self.use_ctc_loss:
                loss_ctc = torch.zeros_like(gate_loss)
            return loss, gate_loss, loss_ctc
        loss = ops.NLLFn.apply(z, out32, float(self.sigma), *log_s_list)
        gate_loss = torch.zeros(1, device=z.device)
        if self.gate_loss > 0:
            gate_loss = ops.GateBCEFn.apply(gate_pred, gate_target, out32)
        loss_ctc = torch.zeros_like(gate_loss)
        if self.use_ctc_loss:
            lps = []
            for i, lp in enumerate(attn_logprob_list):
                if i % 2 != 0:
                    lp = ops.reverse_by_length(lp, out32, False)       # back-step flows are in reversed time (:250-256)
                lps.append(lp)
            # the DP is one workgroup per sample and latency-bound in T: F flows stacked along the batch cost the time
            # of one (F*B of 256 CUs busy).  mean over the F*B stacked samples == mean over flows of the per-flow batch means.
            F_ = len(lps)
            loss_ctc = self.attention_loss(torch.cat(lps, 0) if F_ > 1 else lps[0], in_lengths.repeat(F_), out_lengths.repeat(F_))
        return loss, gate_loss, loss_ctc
