"""Deterministic synthetic weights and batches -- TEST INFRASTRUCTURE ONLY.

numpy's legacy RandomState is bit-stable across numpy versions, so the golden
generator (run once in the build container against the real reference) and
the tests / bench (run anywhere) reconstruct identical inputs from a seed
without shipping 244 MB state_dicts.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

DEFAULT_MODEL_CONFIG = {   # /root/reference/config.json:49-66
    "n_speakers": 1, "n_speaker_dim": 128, "n_text": 185, "n_text_dim": 512,
    "n_flows": 2, "n_mel_channels": 80, "n_attn_channels": 640, "n_hidden": 1024,
    "n_lstm_layers": 2, "mel_encoder_n_hidden": 512, "n_components": 0,
    "mean_scale": 0.0, "fixed_gaussian": True, "dummy_speaker_embedding": False,
    "use_gate_layer": True, "use_cumm_attention": False,
}

SMALL_MODEL_CONFIG = dict(DEFAULT_MODEL_CONFIG, n_speakers=3, n_speaker_dim=16, n_text=40,
                          n_text_dim=32, n_attn_channels=48, n_hidden=64)


def state_dict_spec(cfg: dict):
    """(key, shape) in the reference's registration order."""
    S, C, H, A, M = cfg["n_speaker_dim"], cfg["n_text_dim"], cfg["n_hidden"], cfg["n_attn_channels"], cfg["n_mel_channels"]
    E = C + S
    spec = [("speaker_embedding.weight", (cfg["n_speakers"], S)), ("embedding.weight", (cfg["n_text"], C))]
    for i in range(cfg["n_flows"]):
        p = "flows.%d." % i if i % 2 == 0 else "flows.%d.ar_step." % i
        spec += [(p + "conv.weight", (2 * M, H, 1)), (p + "conv.bias", (2 * M,))]
        for l, inp in [(0, H + A)] + [(l_, H) for l_ in range(1, int(cfg.get("n_lstm_layers", 2)))]:
            spec += [(p + "lstm.weight_ih_l%d" % l, (4 * H, inp)), (p + "lstm.weight_hh_l%d" % l, (4 * H, H)),
                     (p + "lstm.bias_ih_l%d" % l, (4 * H,)), (p + "lstm.bias_hh_l%d" % l, (4 * H,))]
        spec += [(p + "attention_lstm.weight_ih_l0", (4 * H, M)), (p + "attention_lstm.weight_hh_l0", (4 * H, H)),
                 (p + "attention_lstm.bias_ih_l0", (4 * H,)), (p + "attention_lstm.bias_hh_l0", (4 * H,))]
        spec += [(p + "attention_layer.query.linear_layer.weight", (A, H)),
                 (p + "attention_layer.key.linear_layer.weight", (A, E)),
                 (p + "attention_layer.value.linear_layer.weight", (A, E)),
                 (p + "attention_layer.v.linear_layer.weight", (1, A))]
        if cfg.get("use_cumm_attention", False):      # flowtron.py:658-662; nn.Sequential re-registers the two convs
            for nm in ("location_conv_hidden", "location_conv_out", "conv_layers.0", "conv_layers.2"):
                shp = (32, 2, 5) if nm in ("location_conv_hidden", "conv_layers.0") else (E, 32, 3)
                spec += [(p + "attn_cond_layer.%s.conv.weight" % nm, shp), (p + "attn_cond_layer.%s.conv.bias" % nm, (shp[0],))]
        for j in range(2):
            spec += [(p + "dense_layer.layers.%d.linear_layer.weight" % j, (H, H)),
                     (p + "dense_layer.layers.%d.linear_layer.bias" % j, (H,))]
        if i == cfg["n_flows"] - 1 and cfg["use_gate_layer"]:
            spec += [(p + "gate_layer.linear_layer.weight", (1, H + A)), (p + "gate_layer.linear_layer.bias", (1,))]
    for i in range(3):
        spec += [("encoder.convolutions.%d.0.conv.weight" % i, (C, C, 5)), ("encoder.convolutions.%d.0.conv.bias" % i, (C,)),
                 ("encoder.convolutions.%d.1.weight" % i, (C,)), ("encoder.convolutions.%d.1.bias" % i, (C,))]
    Hh = C // 2                                   # BiLSTM hidden per direction
    for sfx in ("", "_reverse"):
        spec += [("encoder.lstm.weight_ih_l0" + sfx, (4 * Hh, C)), ("encoder.lstm.weight_hh_l0" + sfx, (4 * Hh, Hh)),
                 ("encoder.lstm.bias_ih_l0" + sfx, (4 * Hh,)), ("encoder.lstm.bias_hh_l0" + sfx, (4 * Hh,))]
    return spec


def make_state_dict(cfg: dict, seed: int = 1234, coupling_scale: float = 0.02) -> "OrderedDict[str, torch.Tensor]":
    """Random but well-conditioned weights. The coupling conv is NOT zero
    (the reference zero-inits it, flowtron.py:651-653, which would make
    z == mel and test nothing)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for k, shp in state_dict_spec(cfg):
        if k.endswith("embedding.weight"):
            w = rs.standard_normal(shp)
        elif ".1.weight" in k and "convolutions" in k:      # instance-norm gamma
            w = 1.0 + 0.1 * rs.standard_normal(shp)
        elif k.endswith("bias"):
            w = 0.05 * rs.standard_normal(shp)
            if "conv.bias" in k and "convolutions" not in k:
                w = coupling_scale * rs.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            w = rs.uniform(-1.0, 1.0, shp) / np.sqrt(fan_in)
            if "conv.weight" in k and "convolutions" not in k:
                w = w * (coupling_scale * 10)
            if "attention_layer.v." in k:
                w = w * 4.0
        sd[k] = torch.from_numpy(np.ascontiguousarray(w)).float()
    for k in list(sd):                                # aliases of the same module (nn.Sequential view) share values
        if ".conv_layers.0." in k:
            sd[k] = sd[k.replace("conv_layers.0", "location_conv_hidden")].clone()
        elif ".conv_layers.2." in k:
            sd[k] = sd[k.replace("conv_layers.2", "location_conv_out")].clone()
    return sd


def make_batch(cfg: dict, out_lens, in_lens, seed: int = 1234, with_prior: bool = True, n_speakers=None):
    """Synthetic LJS-shape batch. in_lens must be sorted descending
    (data.py:200-202). mel values ~ log-mel range [-11.5, 1]."""
    from .flowtron_oracle import beta_binomial_prior
    rs = np.random.RandomState(seed + 1)
    B = len(out_lens)
    T, L, M = int(max(out_lens)), int(max(in_lens)), cfg["n_mel_channels"]
    mel = np.zeros((B, M, T), np.float32)
    text = np.zeros((B, L), np.int64)
    gate = np.zeros((B, T), np.float32)
    prior = np.zeros((B, T, L), np.float32) if with_prior else None
    for b in range(B):
        t, l = int(out_lens[b]), int(in_lens[b])
        base = -5.0 + 2.0 * np.sin(np.linspace(0, 6.0, M))[:, None]
        walk = np.cumsum(0.15 * rs.standard_normal((1, t)), axis=1)
        mel[b, :, :t] = np.clip(base + walk + 1.2 * rs.standard_normal((M, t)), -11.5, 1.0)
        text[b, :l] = rs.randint(0, cfg["n_text"], size=l)
        gate[b, t - 1:] = 1.0
        if with_prior:
            prior[b, :t, :l] = beta_binomial_prior(l, t).float().numpy()
    nspk = cfg["n_speakers"] if n_speakers is None else n_speakers
    spk = rs.randint(0, nspk, size=B).astype(np.int64)
    out = dict(mel=torch.from_numpy(mel), speaker_ids=torch.from_numpy(spk), text=torch.from_numpy(text),
               in_lens=torch.tensor(list(in_lens), dtype=torch.long), out_lens=torch.tensor(list(out_lens), dtype=torch.long),
               gate_target=torch.from_numpy(gate), attn_prior=torch.from_numpy(prior) if with_prior else None)
    return out


def ljs_like_lengths(B: int, seed: int = 1234, t_max: int = 862, l_max: int = 187):
    """cfg-2 style lengths: out ~ clip(N(566,190),100,862),
    in ~ round(out/5.5) clipped to [12,l_max]; sorted by in_len desc."""
    rs = np.random.RandomState(seed + 7)
    out = np.clip(np.round(rs.normal(566, 190, B)), 100, t_max).astype(int)
    out[0] = t_max
    inn = np.clip(np.round(out / 5.5), 12, l_max).astype(int)
    order = np.argsort(-inn, kind="stable")
    return out[order].tolist(), inn[order].tolist()


def make_audio(n_samples: int, seed: int = 0) -> torch.Tensor:
    rs = np.random.RandomState(seed + 99)
    t = np.arange(n_samples) / 22050.0
    y = np.zeros(n_samples)
    for _ in range(rs.randint(5, 11)):
        y += rs.uniform(0.2, 1.0) * np.sin(2 * np.pi * rs.uniform(80, 4000) * t + rs.uniform(0, 6.28))
    y += 0.05 * rs.standard_normal(n_samples)
    y = 0.9 * y / np.abs(y).max()
    return torch.from_numpy(y.astype(np.float32))
