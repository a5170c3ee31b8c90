"""ctypes binding of libflowtron_hip.so (include/flowtron_hip.h).

This is the stub a maintainer of the reference would add to bind the C ABI
(see INTEGRATION.md).  There is NO fallback: if the shared object is missing
or an entry point fails, a RuntimeError is raised -- the product path never
routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflowtron_hip.so")

FT_F32, FT_BF16, FT_F16 = 0, 1, 2
GEMM_SPLITK = 1
GEMM_SPLITK_DET = 2
GEMM_C16 = 4
ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3

_p, _i, _l, _f, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double


class GemmArgs(C.Structure):
    _fields_ = [("A", _p), ("B", _p), ("C", _p), ("bias", _p),
                ("M", _i), ("N", _i), ("K", _i), ("batch", _i),
                ("sAm", _l), ("sAk", _l), ("sBk", _l), ("sBn", _l), ("ldc", _l),
                ("bsA", _l), ("bsB", _l), ("bsC", _l),
                ("alpha", _f), ("beta", _f), ("act", _i), ("mode", _i), ("flags", _i),
                ("work", _p), ("work_bytes", _sz)]


class GemmImgArgs(C.Structure):
    _fields_ = [("A", _p), ("B", _p), ("C", _p), ("bias", _p), ("M", _i), ("N", _i), ("K", _i),
                ("lda", _l), ("ldb", _l), ("ldc", _l), ("a_kmajor", _i), ("b_kmajor", _i),
                ("alpha", _f), ("beta", _f), ("act", _i), ("flags", _i),
                ("rowmap", _p), ("rows_dev", _p), ("compact", _i), ("k_shift", _i), ("r1_row", _p), ("r1_col", _p),
                ("split_work", _p), ("split_work_bytes", _sz)]


class LstmFwdRole(C.Structure):
    _fields_ = [("gx", _p), ("lens", _p), ("y", _p), ("ldy", _l), ("gates", _p), ("cell", _p), ("wimg", _p), ("state_h", _p), ("state_c", _p),
                ("B", C.c_int32), ("ldb", C.c_int32), ("t0", C.c_int32), ("t1", C.c_int32), ("gx16", C.c_int32)]


class LstmBwdRole(C.Structure):
    _fields_ = [("dy", _p), ("ldy", _l), ("lens", _p), ("gates", _p), ("cell", _p), ("dgx", _p), ("wimg", _p),
                ("dimg", _p), ("dimg_ld", _l), ("dimg_rows", _l), ("dbias", _p), ("state_da", _p), ("state_dc", _p),
                ("B", C.c_int32), ("ldb", C.c_int32), ("t0", C.c_int32), ("t1", C.c_int32), ("carry_in", C.c_int32)]


class ImgDesc(C.Structure):
    _fields_ = [("src", _p), ("ld", _l), ("rows", _l), ("cols", _l), ("dst", _p), ("kind", _i)]


class CummAttnArgs(C.Structure):
    _fields_ = [(n, _p) for n in ("text", "Q", "V", "w_key", "v", "w1", "b1", "w2", "b2", "in_lens", "ctx", "attn", "logprob",
                                  "cumm_all", "kproj_all", "work")] + [("work_bytes", _sz)] + [
        (n, _i) for n in ("T", "B", "L", "E", "A", "NF", "K1", "K2")] + [("temperature", _f), ("mode", _i), ("persist_status", _p)]


class DecodeArgs(C.Structure):
    _fields_ = [(n, _p) for n in (
        "att_w_ih", "att_w_hh", "att_b_ih", "att_b_hh", "w_query", "v", "K", "V",
        "l0_w_ih", "l0_w_hh", "l0_b_ih", "l0_b_hh", "l1_w_ih", "l1_w_hh", "l1_b_ih", "l1_b_hh",
        "d0_w", "d0_b", "d1_w", "d1_b", "conv_w", "conv_b", "gate_w", "gate_b",
        "residual", "mel_out", "attn_out", "n_done_dev", "work")] + [
        ("work_bytes", _sz), ("N", _i), ("L", _i), ("H", _i), ("A", _i), ("M", _i),
        ("temperature", _f), ("gate_threshold", _f), ("use_graph", _i)] + [
        (n, _p) for n in ("cond_w1", "cond_b1", "cond_w2", "cond_b2", "w_key", "enc")] + [("E", _i), ("prior", _p), ("forced", _p),
                                                                                         ("wimg", _p), ("wimg_bytes", _sz), ("persist_gran", _p), ("persist_status", _p),
                                                                                         ("n_layers", _i), ("extra_layers", _p)]


SUMSQ_PARTIALS = 1024      # FT_SUMSQ_PARTIALS: floats of scratch ft_sumsq needs

# name -> argtypes (every symbol include/flowtron_hip.h declares; checked by tests/test_host_cpu.py)
SIGNATURES = {
    "ft_abi_version": ([], _i),
    "ft_last_error": ([], C.c_char_p),
    "ft_debug_hold_cus": ([_i, _l, _p], _i),
    "ft_gemm": ([C.POINTER(GemmArgs), _p], _i),
    "ft_gemm_workspace_bytes": ([C.POINTER(GemmArgs)], _sz),
    "ft_bf16_image_bytes": ([_l, _l], _sz),
    "ft_bf16_image": ([_p, _l, _l, _l, _p, _p], _i),
    "ft_bf16_image_colsum": ([_p, _l, _l, _l, _p, _p, _p], _i),
    "ft_bf16_image_colsum_acc": ([_p, _l, _l, _l, _p, _p, _p], _i),
    "ft_gemm_img": ([C.POINTER(GemmImgArgs), _p], _i),
    "ft_gemm_img_split_work_bytes": ([_i, _i, _i], _sz),
    "ft_bf16_image_split3": ([_p, _l, _l, _l, _p, _i, _p], _i),
    "ft_bf16_image_split3_im2col": ([_p, _p, _i, _i, _i, _i, _p, _p], _i),
    "ft_bf16_image_split3_im2col_f16": ([_p, _p, _i, _i, _i, _i, _p, _p], _i),
    "ft_bf16_image_table": ([C.POINTER(ImgDesc), _i, _p], _i),
    "ft_bf16_image_table_f16": ([C.POINTER(ImgDesc), _i, _p], _i),
    "ft_bf16_image_split3_f16": ([_p, _l, _l, _l, _p, _i, _p], _i),
    "ft_rowmap_build": ([_p, _p, _p, _i, _i, _p], _i),
    "ft_bf16_image_rows": ([_p, _l, _l, _l, _p, _p, _p, _p, _p], _i),
    "ft_bf16_image_rows_acc": ([_p, _l, _l, _l, _p, _p, _p, _p, _p], _i),
    "ft_bf16_image_rows_into": ([_p, _l, _l, _l, _p, _l, _l, _l, _p, _p, _p], _i),
    "ft_img_gemv_rows": ([_p, _l, _i, _p, _p, _p, _l, _p, _p, _p, _i, _i, _p], _i),
    "ft_img_gemv_rows_bwd": ([_p, _l, _i, _p, _l, _p, _p, _p, _p, _l, _p], _i),
    "ft_bf16_image_rows_act_bwd": ([_p, _l, _p, _l, _i, _l, _l, _p, _p, _p, _p, _p], _i),
    "ft_bf16_image_rows_act_bwd_acc": ([_p, _l, _p, _l, _i, _l, _l, _p, _p, _p, _p, _p], _i),
    "ft_pad_rows_fill": ([_p, _l, _i, _p, _i, _i, _i, _p], _i),
    "ft_embedding_fwd": ([_p, _p, _p, _i, _i, _l, _p], _i),
    "ft_embedding_bwd": ([_p, _p, _p, _i, _i, _l, _p], _i),
    "ft_embedding_bwd_runs": ([_p, _p, _p, _i, _i, _l, _i, _p], _i),
    "ft_im2col": ([_p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_col2im": ([_p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_instnorm_relu_fwd": ([_p] * 8 + [_i, _i, _i, _f, _p], _i),
    "ft_instnorm_relu_bwd": ([_p] * 11 + [_i, _i, _i, _p], _i),
    "ft_lstm_workspace_bytes": ([_i, _i], _sz),
    "ft_lstm_seq_fwd": ([_p, _p, _p, _p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "ft_lstm_seq_bwd": ([_p, _l, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "ft_lstm_persist_supported": ([_i, _i], _i),
    "ft_lstm_persist_workspace_bytes": ([_i, _i], _sz),
    "ft_lstm_persist_debug_prof": ([_p], _i),
    "ft_lstm_persist_bwd": ([_p, _l, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_bilstm_persist_supported": ([_i, _i], _i),
    "ft_bilstm_persist_workspace_bytes": ([_i, _i], C.c_size_t),
    "ft_bilstm_persist_fwd": ([_p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "ft_bilstm_persist_bwd": ([_p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "ft_lstm_persist_bwd_img": ([_p, _l, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _l, _l, _p, _p], _i),
    "ft_lstm_roles_ctx_bytes": ([], _sz),
    "ft_lstm_roles_wimg_bytes": ([_i], _sz),
    "ft_lstm_roles_ctx_init": ([_p, _p], _i),
    "ft_lstm_roles_debug_prof": ([_p], _i),
    "ft_lstm_roles_prepare_fwd": ([_p, _p, _i, _p], _i),
    "ft_lstm_roles_prepare_bwd": ([_p, _p, _i, _p], _i),
    "ft_lstm_roles_fwd": ([C.POINTER(LstmFwdRole), _i, _i, _i, _p, _i, _p, _i, _p], _i),
    "ft_lstm_roles_bwd": ([C.POINTER(LstmBwdRole), _i, _i, _i, _p, _i, _p, _i, _p], _i),
    "ft_lstm2_supported": ([_i, _i], _i),
    "ft_lstm2_workspace_bytes": ([_i, _i], _sz),
    "ft_lstm2_seq_fwd": ([_p] * 13 + [_i, _i, _i, _p], _i),
    "ft_lstm2_seq_bwd": ([_p] * 12 + [_i, _i, _i, _p], _i),
    "ft_lstm_bidir_supported": ([_i, _i], _i),
    "ft_lstm_bidir_seq_fwd": ([_p] * 6 + [_l] + [_p] * 6 + [_i, _i, _i, _p], _i),
    "ft_lstm_bidir_seq_bwd": ([_p, _l] + [_p] * 11 + [_i, _i, _i, _p], _i),
    "ft_cumm_attn_workspace_bytes": ([_i] * 10, _sz),
    "ft_cumm_attn_fused": ([C.POINTER(CummAttnArgs)], _i),
    "ft_cumm_attn_debug_prof": ([_p], _i),
    "ft_cumm_attn_fwd": ([C.POINTER(CummAttnArgs), _p], _i),
    "ft_cumm_attn_bwd": ([C.POINTER(CummAttnArgs)] + [_p] * 13, _i),
    "ft_attention_fwd": ([_p] * 8 + [_i, _i, _i, _i, _f, _p], _i),
    "ft_attention_bwd": ([_p] * 13 + [_i, _i, _i, _i, _f, _p], _i),
    "ft_affine_fwd": ([_p, _p, _p, _l, _i, _p], _i),
    "ft_affine_bwd": ([_p, _p, _p, _p, _p, _p, _l, _i, _p], _i),
    "ft_affine_inv": ([_p, _p, _p, _l, _i, _p], _i),
    "ft_masked_sum": ([_p, _l, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_masked_sum_bwd": ([_p, _l, _p, _p, _f, _i, _p, _l, _i, _i, _i, _p], _i),
    "ft_gate_bce_fwd": ([_p, _p, _p, _p, _i, _i, _p], _i),
    "ft_gate_bce_bwd": ([_p, _p, _p, _p, _f, _p, _i, _i, _p], _i),
    "ft_flowtron_loss_fwd": ([_p, _p, _i, _l, _p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _p], _i),
    "ft_flowtron_loss_bwd": ([_p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "ft_reverse_by_length": ([_p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_act_bwd": ([_p, _p, _p, _l, _i, _p], _i),
    "ft_eltwise": ([_p, _p, _p, _l, _i, _p], _i),
    "ft_colsum": ([_p, _p, _l, _i, _l, _p], _i),
    "ft_decode_workspace_bytes": ([_i, _i, _i, _i, _i, _i], _sz),
    "ft_decode_wimg_bytes": ([_i, _i, _i], _sz),
    "ft_decode_persist_gran_bytes": ([], _sz),
    "ft_decode_debug_prof": ([_p], _i),
    "ft_decode_flow": ([C.POINTER(DecodeArgs), _p], _i),
    "ft_stft_mel": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "ft_stft_r8": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_stft_r8_ragged": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "ft_attn_ctc_workspace_floats": ([_i, _i, _i], _sz),
    "ft_attn_ctc_fwd": ([_p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_attn_ctc_bwd": ([_p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_attn_ctc_fwd_multi": ([_p, _p, _i, _p, _p, _f, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_attn_ctc_bwd_multi": ([_p, _p, _i, _p, _p, _f, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "ft_beta_binomial_prior": ([_p, _p, _p, _i, _i, _i, _f, _p], _i),
    "ft_sumsq": ([_p, _p, _l, _p, _p], _i),
    "ft_radam_step": ([_p, _p, _p, _p, _l, _p, _d, _d, _d, _d, _d, _d, _d, _i, _p, _p], _i),
    "ft_radam_step_dev": ([_p, _p, _p, _p, _l, _p, _d, _d, _d, _d, _d, _d, _i, _p, _p], _i),
    "ft_poison_if_nonzero": ([_p, _p, _p], _i),
}

# fp16-operand twins (include/flowtron_hip.h, end): same signatures, suffix _f16
OP16_TWINS = ("ft_gemm", "ft_bf16_image", "ft_bf16_image_colsum", "ft_bf16_image_colsum_acc", "ft_bf16_image_rows_acc", "ft_bf16_image_rows_act_bwd_acc", "ft_gemm_img", "ft_bf16_image_rows", "ft_bf16_image_rows_into", "ft_bf16_image_rows_act_bwd", "ft_img_gemv_rows", "ft_img_gemv_rows_bwd", "ft_lstm_seq_fwd", "ft_lstm_seq_bwd",
              "ft_lstm_persist_bwd", "ft_lstm_persist_bwd_img", "ft_lstm_roles_prepare_fwd", "ft_lstm_roles_prepare_bwd", "ft_lstm_roles_fwd", "ft_lstm_roles_bwd", "ft_lstm2_seq_fwd", "ft_lstm2_seq_bwd",
              "ft_lstm_bidir_seq_fwd", "ft_lstm_bidir_seq_bwd", "ft_bilstm_persist_fwd", "ft_bilstm_persist_bwd")
for _n in OP16_TWINS:
    SIGNATURES[_n + "_f16"] = SIGNATURES[_n]

_lib = None


def lib():
    """Loads the shared object once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libflowtron_hip.so is missing (%s). Build it with `python -m flowtron_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (argt, rest) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argt
            fn.restype = rest
        if l.ft_abi_version() != 14:
            raise RuntimeError("libflowtron_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().ft_last_error().decode()))


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "flowtron_amd ops run on MI355X only: got a %s tensor. There is no CPU fallback "
                "(the CPU oracle lives in oracle/ and is test infrastructure)." % t.device)


def mfma_mode() -> int:
    """Operand type fed to the matrix cores.  Storage, accumulation and every non-matmul kernel are fp32 in all modes.
    FLOWTRON_MFMA = f32 | bf16 | f16 pins it; unset (or "auto") follows the caller's torch.autocast region the way the
    reference's AMP does (train.py:292 `with amp.autocast(enabled=fp16_run)`): float16 -> FT_F16, bfloat16 -> FT_BF16,
    no autocast -> FT_F32.  Autograd nodes record the mode at forward time; backward never re-reads it."""
    m = os.environ.get("FLOWTRON_MFMA", "auto").lower()
    if m == "auto":
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            return FT_F16 if dt == torch.float16 else FT_BF16
        return FT_F32
    if m in ("f32", "fp32"):
        return FT_F32
    if m in ("bf16",):
        return FT_BF16
    if m in ("f16", "fp16"):
        return FT_F16
    raise ValueError("FLOWTRON_MFMA must be auto, f32, bf16 or f16, got %r" % m)


def is16(mode: int) -> bool:
    """16-bit operand modes: the image / fragment / persistent-recurrence path."""
    return mode in (FT_BF16, FT_F16)


def op16(name: str, mode: int):
    """The entry point `name` of the 16-bit operand format `mode` (bf16: the plain name, fp16: its _f16 twin)."""
    if mode == FT_F16:
        return getattr(lib(), name + "_f16")
    if mode == FT_BF16:
        return getattr(lib(), name)
    raise ValueError("%s needs a 16-bit operand mode, got %d" % (name, mode))
