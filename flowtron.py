"""Drop-in replacement for the reference's `flowtron.py` module (NVIDIA/flowtron):

    from flowtron import Flowtron, FlowtronLoss          # train.py:25-26, inference.py:29

Same class names, constructor arguments (config.json model_config keys), submodule names /
state_dict keys, and Flowtron.forward / Flowtron.infer signatures; checkpoints that pickle
`flowtron.Flowtron` (train.py:131-139) load against this module.  The implementation lives in
flowtron_amd/ (HIP kernels behind the C ABI in include/flowtron_hip.h); there is no CPU path.
"""
import torch

from flowtron_amd.model import (AR_Back_Step, AR_Step, Attention, AttentionConditioningLayer, AttentionCTCLoss, ConvNorm,
                                DenseLayer, Encoder,
                                Flowtron, FlowtronLoss, LinearNorm, MaskedInstanceNorm1d)

for _cls in (AR_Back_Step, AR_Step, Attention, AttentionConditioningLayer, AttentionCTCLoss, ConvNorm, DenseLayer, Encoder, Flowtron,
             FlowtronLoss, LinearNorm, MaskedInstanceNorm1d):
    _cls.__module__ = "flowtron"          # pickled checkpoints resolve `flowtron.<Class>`


def get_mask_from_lengths(lengths):
    """bool [B, max_len], True at valid positions (flowtron.py:39-50), device-agnostic and sync-free
    when max_len is known from a tensor shape by the caller."""
    max_len = int(lengths.max())
    ids = torch.arange(0, max_len, device=lengths.device, dtype=lengths.dtype)
    return ids[None, :] < lengths[:, None]


def get_gate_mask_from_lengths(lengths):
    """(flowtron.py:25-36) identical mask, kept for API parity."""
    return get_mask_from_lengths(lengths)
