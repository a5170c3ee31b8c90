"""torch.autograd.Function wrappers around the C ABI (include/flowtron_hip.h).

Each Function's forward AND backward enqueue hand-written HIP kernels on torch's
current stream through ctypes; torch supplies device memory and the autograd
graph only.  Nothing here computes on the CPU and nothing falls back to torch
ops for the math: a missing library or a CPU tensor raises.

Layouts (reference flowtron.py internal layout after :884): activations are
time-major [T,B,C] fp32 contiguous; lengths are int32 device tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L

# --------------------------------------------------------------------------
# raw launchers
# --------------------------------------------------------------------------


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


import os as _os
_BF16_IMAGES = True        # large 16-bit GEMMs run from shared operand images (round 1; the switch was retired in round 6)


def gemm_raw(A, B, Cm, M, N, K, sAm, sAk, sBk, sBn, ldc, bias=None, act=L.ACT_NONE, alpha=1.0, beta=0.0,
             batch=1, bsA=0, bsB=0, bsC=0, mode=None, splitk=False):
    """C = act(alpha*A.B + beta*C + bias); A,B,Cm are tensors whose data_ptr is the operand origin.
    splitk=True (weight-gradient GEMMs only) allows the atomic split-K path."""
    L.require_cuda(A, B, Cm, bias)
    a = L.GemmArgs(L.ptr(A), L.ptr(B), L.ptr(Cm), L.ptr(bias), M, N, K, batch,
                   sAm, sAk, sBk, sBn, ldc, bsA, bsB, bsC, alpha, beta, act,
                   L.mfma_mode() if mode is None else mode, L.GEMM_SPLITK if splitk else 0,
                   None, 0)
    if _BF16_IMAGES:
        need = L.lib().ft_gemm_workspace_bytes(C.byref(a))
        if need:                      # large bf16-mode GEMM: bf16 operand images + DMA-staged kernel (gemm_bf16.hip)
            work = torch.empty(need, device=Cm.device, dtype=torch.uint8)     # stream-ordered by the caching allocator
            a.work, a.work_bytes = L.ptr(work), need
    L.check(L.lib().ft_gemm(C.byref(a), L.stream()), "ft_gemm")


# --------------------------------------------------------------------------
# bf16 operand images (csrc/gemm_bf16.hip): every fp32 matrix that feeds a large bf16-mode GEMM is rounded ONCE and the
# image serves all its GEMMs -- x: forward + dW, dY: dX + dW, W: forward + dX, LSTM dgates: dW_hh + dW_ih (+ dX) --
# k-contiguous or k-major as the GEMM needs (k-major fragments come from the LDS transpose-read, so nothing is transposed).
# --------------------------------------------------------------------------
class RowMap:
    """Pack-by-length row map of the time-major decoder activations [T,B,*] (csrc: ft_rowmap_build; the reference packs with
    pack_padded_sequence, flowtron.py:689-694): compact rows are batch-major, utterance b = its valid frames + ONE separator
    (its first padded frame, or a zero row when it has none).  Built on the device from the int32 length vector -- no host
    sync; `cap` = T*B + B is the capacity images and grids are sized for, `rows` the device-side row count."""
    __slots__ = ("T", "B", "cap", "lens", "map", "rows")

    def __init__(self, lens32, T, B):
        L.require_cuda(lens32)
        self.T, self.B, self.cap, self.lens = int(T), int(B), int(T) * int(B) + int(B), lens32
        self.map = torch.empty(self.cap, device=lens32.device, dtype=torch.int32)
        self.rows = torch.empty(1, device=lens32.device, dtype=torch.int32)
        L.check(L.lib().ft_rowmap_build(L.ptr(lens32), L.ptr(self.map), L.ptr(self.rows), self.T, self.B, L.stream()), "ft_rowmap_build")

    def fill(self, y2d, ncols, copy_separator):
        """padded frames t > len_b of the time-major matrix y2d [T*B, >= ncols]: zeros, or (copy_separator) the values of the
        utterance's first padded frame, which a compact GEMM has just computed"""
        L.check(L.lib().ft_pad_rows_fill(L.ptr(y2d), int(y2d.stride(0)), int(ncols), L.ptr(self.lens), self.T, self.B,
                                         1 if copy_separator else 0, L.stream()), "ft_pad_rows_fill")


_COMPACT = _os.environ.get("FLOWTRON_GEMM_COMPACT", "1") != "0"
_FUSE_ACT_BWD = True       # activation backward inside the gradient's image pass (round 4)
# split-K on the encoder's FORWARD split-image GEMM (160 output tiles, K = 7 680).  With fp32 atomics (the first form: the
# first form, -0.2 ms per step) the order of the atomics makes the forward itself differ from run to run (z up to 8e-4, bf16 gradients
# up to 1e-2 rel-L2 on the same batch and weights: scripts/exp/noise_debug.py, profiles/r05b_forward_noise.log).  Default "det": the
# slices' partial products side by side + a fixed-order reduction (FT_GEMM_SPLITK_DET) -- a forward pass is a function of its inputs;
# "0": one slice
_ENC_SPLITK = "det"
_CAT_IMAGES = True         # a Linear over two inputs as ONE GEMM over a concatenated image (round 4)
_PERSIST_IMG = _os.environ.get("FLOWTRON_LSTM_PERSIST_IMG", "1")      # 1: the persistent backward emits the dgates image INSTEAD of fp32 dgx where its only consumer is the projection's backward; both; 0


def row_map(lens32, T, B):
    """the RowMap the 16-bit image GEMMs of a [T,B,*] stack should use, or None (FLOWTRON_GEMM_COMPACT=0: padded rows are multiplied)"""
    return RowMap(lens32, T, B) if (_COMPACT and _BF16_IMAGES) else None


class Bf16Image:
    __slots__ = ("buf", "rows", "cols", "ld", "colsum", "fmt", "rowmap")

    def __init__(self, t2d, colsum=False, mode=None, rowmap=None):
        """t2d: fp32 CUDA matrix [rows, cols], unit column stride.  colsum=True also returns the fp32 column sums of t2d
        (self.colsum) from the same pass -- the bias gradient when t2d is an output gradient.
        rowmap: a RowMap over t2d's rows (t2d = a time-major [T*B, cols] activation): the image holds the VALID rows only, in
        compact order (self.rows = the map's capacity)."""
        assert t2d.dim() == 2 and t2d.stride(1) == 1 and t2d.dtype == torch.float32
        L.require_cuda(t2d)
        self.rows, self.cols = int(t2d.shape[0]), int(t2d.shape[1])
        self.fmt = L.mfma_mode() if mode is None else mode      # FT_BF16 or FT_F16: 16-bit payloads of that operand format
        self.ld = (self.cols + 255) // 256 * 256
        self.rowmap = rowmap
        # column sums (bias gradients) are ADDED to a zeroed slice of the backward pass's slab (the _acc entry points: no memset each)
        self.colsum = zeroed((self.cols,), t2d.device) if colsum else None
        if rowmap is not None:
            assert self.rows == rowmap.T * rowmap.B, "row map built for another [T, B]"
            self.rows = rowmap.cap
            self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=t2d.device, dtype=torch.uint8)
            L.check(L.op16("ft_bf16_image_rows_acc", self.fmt)(L.ptr(t2d), int(t2d.stride(0)), self.rows, self.cols, L.ptr(self.buf), L.ptr(self.colsum),
                                                                L.ptr(rowmap.map), L.ptr(rowmap.rows), L.stream()), "ft_bf16_image_rows_acc")
            return
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=t2d.device, dtype=torch.uint8)
        if colsum:
            L.check(L.op16("ft_bf16_image_colsum_acc", self.fmt)(L.ptr(t2d), int(t2d.stride(0)), self.rows, self.cols, L.ptr(self.buf), L.ptr(self.colsum),
                                                     L.stream()), "ft_bf16_image_colsum_acc")
        else:
            L.check(L.op16("ft_bf16_image", self.fmt)(L.ptr(t2d), int(t2d.stride(0)), self.rows, self.cols, L.ptr(self.buf), L.stream()), "ft_bf16_image")

    def ptr(self, row_off=0, col_off=0):
        assert col_off % 8 == 0
        return self.buf.data_ptr() + 2 * (row_off * self.ld + col_off)

    def view_cols(self, cols):
        """the first `cols` columns as an image of their own (same storage and row stride): the [hi] block of a split image IS the
        plain image of the matrix"""
        v = Bf16Image.__new__(Bf16Image)
        v.buf, v.rows, v.cols, v.ld, v.colsum, v.fmt, v.rowmap = self.buf, self.rows, int(cols), self.ld, None, self.fmt, self.rowmap
        return v

    @classmethod
    def split3(cls, t2d, mode, weight):
        """[hi | lo | hi] (activation) / [hi | hi | lo] (weight) image of t2d [rows, K]: one 16-bit GEMM over 3 K columns then gives
        fp32-grade products (ft_bf16_image_split3)"""
        assert t2d.dim() == 2 and t2d.stride(1) == 1 and t2d.dtype == torch.float32 and t2d.shape[1] % 8 == 0
        self = cls.__new__(cls)
        self.rows, self.cols, self.fmt, self.rowmap, self.colsum = int(t2d.shape[0]), 3 * int(t2d.shape[1]), mode, None, None
        self.ld = (self.cols + 255) // 256 * 256
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=t2d.device, dtype=torch.uint8)
        L.check(L.op16("ft_bf16_image_split3", mode)(L.ptr(t2d), int(t2d.stride(0)), self.rows, int(t2d.shape[1]), L.ptr(self.buf),
                                                     1 if weight else 0, L.stream()), "ft_bf16_image_split3")
        return self

    @classmethod
    def split3_im2col(cls, x, lens, KW, mode):
        """the [hi | lo | hi] image of ft_im2col(x, lens, KW) -- x [L, B, C] -- made straight from x (ft_bf16_image_split3_im2col): the
        fp32 column matrix is never written"""
        Lx, B, Cc = x.shape
        self = cls.__new__(cls)
        self.rows, self.cols, self.fmt, self.rowmap, self.colsum = int(Lx * B), 3 * int(Cc * KW), mode, None, None
        self.ld = (self.cols + 255) // 256 * 256
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=x.device, dtype=torch.uint8)
        L.check(L.op16("ft_bf16_image_split3_im2col", mode)(L.ptr(x), L.ptr(lens), int(Lx), int(B), int(Cc), int(KW), L.ptr(self.buf), L.stream()),
                "ft_bf16_image_split3_im2col")
        return self

    @classmethod
    def _blank(cls, rows, cols, mode, device):
        self = cls.__new__(cls)
        self.rows, self.cols, self.fmt, self.rowmap, self.colsum = int(rows), int(cols), mode, None, None
        self.ld = (self.cols + 255) // 256 * 256
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=device, dtype=torch.uint8)
        return self

    @classmethod
    def of_weight(cls, W, mode, split=False):
        """the (plain | split [hi | hi | lo]) image of a WEIGHT matrix.  Inside a Flowtron.forward (weight_images_begin) the images of
        all weights the previous forward of that model asked for are rounded in ONE launch at the first request
        (ft_bf16_image_table: 23 dispatches of 5-15 us -> 1) -- from the CURRENT values, every forward anew: nothing is cached across
        steps (a weight changed through `.data` would not bump its version).  Outside, or for a weight not in the plan: one launch."""
        w = _WIMG
        key = (W.data_ptr(), tuple(W.shape), mode, bool(split))
        if w["plan_next"] is not None:
            w["plan_next"][key] = W
        if w["armed"]:
            w["armed"] = False
            # (a planned tensor whose storage moved since the previous forward -- FlatArena flattening after the first forward, a p.data
            #  swap -- or changed shape is dropped: its image would sit under a stale key that no request finds, or that a same-shape
            #  tensor reusing the old address could pick up)
            plan = [(k, t) for k, t in w["plan"].items() if t.is_cuda and t.device == W.device and t.dtype == torch.float32 and t.is_contiguous()
                    and t.data_ptr() == k[0] and tuple(t.shape) == k[1]]
            for lo in range(0, len(plan), 32):
                part = plan[lo:lo + 32]
                by_fmt = {}
                for k, t in part:
                    by_fmt.setdefault(k[2], []).append((k, t))
                for fmt, items in by_fmt.items():
                    descs = (L.ImgDesc * len(items))()
                    for i, (k, t) in enumerate(items):
                        img = cls._blank(t.shape[0], (3 if k[3] else 1) * t.shape[1], fmt, t.device)
                        descs[i] = L.ImgDesc(t.data_ptr(), int(t.stride(0)), int(t.shape[0]), int(t.shape[1]), img.buf.data_ptr(), 1 if k[3] else 0)
                        w["cache"][k] = img
                    L.check(L.op16("ft_bf16_image_table", fmt)(descs, len(items), L.stream()), "ft_bf16_image_table")
        img = w["cache"].get(key)
        if img is not None:
            return img
        return cls.split3(W, mode, True) if split else cls(W, mode=mode)

    @classmethod
    def cat_rows(cls, xs2d, mode, rowmap):
        """ONE compact image of the column-wise concatenation [x_0 | x_1 | ..] of time-major activations (each [T*B, K_i] fp32,
        K_i % 8 == 0 except the last): a Linear over several inputs then runs as one GEMM with one K loop (flowtron.py:758-765
        concatenates [h_att ; ctx] before the decoder LSTM), and its weight gradient as one split-K GEMM."""
        self = cls.__new__(cls)
        self.rows, self.cols, self.fmt, self.rowmap, self.colsum = rowmap.cap, int(sum(x.shape[1] for x in xs2d)), mode, rowmap, None
        self.ld = (self.cols + 255) // 256 * 256
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=xs2d[0].device, dtype=torch.uint8)
        off = 0
        for i, x in enumerate(xs2d):
            assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and x.shape[0] == rowmap.T * rowmap.B
            K = int(x.shape[1])
            last = i == len(xs2d) - 1
            assert last or K % 8 == 0
            L.check(L.op16("ft_bf16_image_rows_into", mode)(L.ptr(x), int(x.stride(0)), self.rows, K, L.ptr(self.buf), self.ld, off,
                                                            (self.ld - off) if last else K, L.ptr(rowmap.map), L.ptr(rowmap.rows), L.stream()),
                    "ft_bf16_image_rows_into")
            off += K
        return self

    @classmethod
    def empty_rows(cls, cols, rowmap, mode, device):
        """an UNWRITTEN compact image over `rowmap` ([cap, cols] 16-bit payloads of format `mode`) with zeroed column sums, for a
        kernel that produces the image itself (ft_lstm_persist_bwd_img)"""
        self = cls.__new__(cls)
        self.rows, self.cols, self.fmt, self.rowmap = rowmap.cap, int(cols), mode, rowmap
        self.ld = (self.cols + 255) // 256 * 256
        self.buf = torch.empty(L.lib().ft_bf16_image_bytes(self.rows, self.cols), device=device, dtype=torch.uint8)
        self.colsum = zeroed((self.cols,), device)
        return self


# weight images of one forward pass in one launch (Bf16Image.of_weight): the plan = what the previous forward of the same model asked for
_WIMG_ON = True
_WIMG = {"plan": {}, "plan_next": None, "cache": {}, "armed": False, "plans": None}


def weight_images_begin(model):
    """called at the head of Flowtron.forward: arm the one-launch conversion of the weights this model's previous forward used"""
    if not _WIMG_ON:
        return
    w = _WIMG
    if w["plans"] is None:
        w["plans"] = _weakref.WeakKeyDictionary()
    w["plan"] = w["plans"].get(model, {})
    w["plan_next"], w["cache"], w["armed"] = {}, {}, bool(w["plan"])


def weight_images_end(model):
    w = _WIMG
    if w["plan_next"] is not None and w["plans"] is not None:
        w["plans"][model] = w["plan_next"]
    w["plan"], w["plan_next"], w["cache"], w["armed"] = {}, None, {}, False


# images of activations are shared between every consumer of the SAME tensor (h_att feeds the query projection, the gate and
# the decoder input projection; an LSTM output also feeds its own dW_hh GEMM in backward): keyed by tensor identity
# (id + weakref + version), dropped with the tensor.
import weakref as _weakref
_IMG_CACHE = {}


def shared_image(t, rows, cols, mode, rowmap=None):
    """16-bit image (operand format `mode`) of the fp32 tensor `t` viewed as [rows, cols]; one conversion per tensor (version)
    and row map."""
    key = id(t)
    hit = _IMG_CACHE.get(key)
    want_rows = rows if rowmap is None else rowmap.cap
    if hit is not None:
        ref, ver, img = hit
        if ref() is t and ver == t._version and img.rows == want_rows and img.cols == cols and img.fmt == mode and img.rowmap is rowmap:
            return img
    img = Bf16Image(t.reshape(rows, cols), mode=mode, rowmap=rowmap)
    if len(_IMG_CACHE) > 64:                     # dead entries (their tensors are gone) are swept lazily
        for k in [k for k, (r, _, _) in _IMG_CACHE.items() if r() is None]:
            del _IMG_CACHE[k]
    try:
        _IMG_CACHE[key] = (_weakref.ref(t, lambda _r, k=key: _IMG_CACHE.pop(k, None)), t._version, img)
    except TypeError:
        pass
    return img


def op16_dtype(mode):
    """torch dtype of a tensor that holds 16-bit values of the operand format `mode`"""
    return torch.float16 if mode == L.FT_F16 else torch.bfloat16


_GX16 = True     # gx of the persistent forward recurrences as 16-bit rows (round 6; tests set it False to compare with fp32 rows)


def gx16_ok(mode, rowmap, T, B, H, reverse, xs, N, device):
    """whether an input projection may write its output as 16-BIT rows (FT_GEMM_C16): its only reader is a persistent forward recurrence
    (csrc/lstm_roles.hip takes them) and its gradient travels back as the dgates image alone (so the fp32 face autograd would
    otherwise round to 16 bits never carries values)"""
    return (_GX16 and L.is16(mode) and rowmap is not None and not reverse and rowmap.T == T and rowmap.B == B and _PERSIST_IMG == "1"
            and linear_uses_images(mode, T * B, N, xs) and (len(xs) == 1 or _CAT_IMAGES)
            and bool(lstm_persist_groups(B, H, reverse, mode, device) or lstm_persist_slices(B, H, reverse, mode, device))
            and B <= 32)


def images_apply(mode, M, N, K):
    """same rule as ft_gemm_workspace_bytes: bf16 mode and a GEMM large enough to amortise the image passes."""
    return _BF16_IMAGES and L.is16(mode) and M >= 32 and N >= 32 and K >= 16 and M * N * K >= (1 << 20)


def gemm_img(A, a_km, a_ptr, B, b_km, b_ptr, Cm, M, N, K, ldc, bias=None, act=L.ACT_NONE, alpha=1.0, beta=0.0, splitk=False,
             rowmap=None, compact=0, k_shift=0, rank1=None, c16=False):
    """C[M,N] = act(alpha * A.B + beta*C + bias) from images.  a_ptr / b_ptr: A.ptr(...) / B.ptr(...) (may point inside).
    rowmap + compact: 1 = M runs over the map's compact rows (pass M = rowmap.cap), C rows are scattered through the map;
    2 = the reduction runs over compact rows (pass K = rowmap.cap; k_shift = a row shift already applied to a_ptr).
    rank1 = (r [C rows] fp32, c_ptr -> N floats): C[row][col] += r[row] * c[col] in the epilogue (row = the output row)."""
    L.require_cuda(Cm, bias)
    flags, work, need = (L.GEMM_SPLITK if splitk else 0), None, 0
    if splitk == "det":
        # deterministic split-K (forward GEMMs): partial products side by side in a workspace, added in a fixed order
        need = L.lib().ft_gemm_img_split_work_bytes(M, N, K)
        flags = L.GEMM_SPLITK_DET if need else 0
        work = torch.empty(need, device=Cm.device, dtype=torch.uint8) if need else None
    if c16:                                      # C is a 16-bit tensor of the operands' format (FT_GEMM_C16)
        assert Cm.dtype == op16_dtype(A.fmt) and not splitk and beta == 0.0
        flags |= L.GEMM_C16
    a = L.GemmImgArgs(a_ptr, b_ptr, L.ptr(Cm), L.ptr(bias), M, N, K, A.ld, B.ld, ldc, int(a_km), int(b_km),
                      alpha, beta, act, flags,
                      L.ptr(rowmap.map) if rowmap is not None else None, L.ptr(rowmap.rows) if rowmap is not None else None,
                      int(compact) if rowmap is not None else 0, int(k_shift),
                      L.ptr(rank1[0]) if rank1 is not None else None, rank1[1] if rank1 is not None else None,
                      L.ptr(work), need)
    assert A.fmt == B.fmt, "operand images of different formats"
    L.check(L.op16("ft_gemm_img", A.fmt)(C.byref(a), L.stream()), "ft_gemm_img")


# hand-off of an output-gradient image between two autograd nodes of the SAME backward pass (the LSTM backward builds the
# image of dgates for its own dW_hh GEMM; the input projection's LinearFn.backward receives that very tensor as dy).
# Keyed by (device, data_ptr, shape); cleared by an engine callback at the end of the pass, so an address can never match stale data.
_HANDOFF = {"imgs": {}, "armed": False, "unwritten": set(), "nan_next": 0}
_NAN_POOL = {}          # device -> 64 fp32 NaNs: one slot per image-only gradient alive in a backward pass


def _current_graph_task():
    try:
        return torch._C._current_graph_task_id()
    except Exception:                   # older torch: fall back to "one pass at a time" (the pre-round-5 behaviour)
        return -1


def _handoff_clear():
    _HANDOFF["imgs"].clear()
    _HANDOFF["unwritten"].clear()
    _HANDOFF["armed"] = False
    _HANDOFF["nan_next"] = 0


def image_only_gradient(shape, device, dtype=torch.float32):
    """The fp32 face of a gradient that exists ONLY as a 16-bit operand image (ft_lstm_persist_bwd_img): every element reads NaN.
    No storage of the gradient's size is allocated -- ONE NaN float, expanded with zero strides; each such gradient of a backward
    pass takes its own slot of a small pool, so (device, data_ptr, shape) still identifies it for the image hand-off.  A foreign
    reader (a node hook, anomaly mode's output check, an unexpected second consumer) therefore sees NaN -- never stale memory."""
    pool = _NAN_POOL.get(device)
    if pool is None:
        pool = _NAN_POOL[device] = torch.full((64,), float("nan"), device=device, dtype=torch.float32)
    i = _HANDOFF["nan_next"]
    _HANDOFF["nan_next"] = i + 1
    if dtype != torch.float32:
        # (a 16-bit face -- the gradient of 16-bit gx rows: its own pool of that dtype, the slot index keeps the address unique)
        pool = _NAN_POOL.get((device, dtype))
        if pool is None:
            pool = _NAN_POOL[(device, dtype)] = torch.full((64,), float("nan"), device=device, dtype=dtype)
    slot = pool[i:i + 1] if i < 64 else torch.full((1,), float("nan"), device=device, dtype=dtype)
    return slot.expand(*shape) if len(shape) else slot.reshape(())


def _handoff_put(t, img, key=None):
    # one clear callback per backward PASS.  The engine drops its callbacks when a backward raises, which used to leave `armed`
    # set for good: imgs / unwritten / nan_next were then never cleared again.  A pass is identified by the engine's
    # current graph task; a put from another pass than the one that armed the state clears the leftovers and arms again.
    gt = _current_graph_task()
    if not _HANDOFF["armed"] or _HANDOFF.get("task") != gt:
        if _HANDOFF["armed"]:
            _handoff_clear()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_handoff_clear)
        except RuntimeError:            # not inside a backward pass: no consumer can follow
            return
        _HANDOFF["armed"] = True
        _HANDOFF["task"] = gt
    _HANDOFF["imgs"][(t.device.index, t.data_ptr(), tuple(t.shape) if key is None else key)] = img


def _handoff_put_x(x, img):
    """the operand image a Linear's backward has just used for its INPUT x, left for the node that produced x: an LSTM's backward
    multiplies the same compact image of its output (dW_hh = dgates^T . y shifted by one row) and would otherwise convert y again"""
    if img is not None and img.rowmap is not None:
        _handoff_put(x, img, key=("x", x.numel()))


def _handoff_take_x(y, cols, mode, rowmap):
    img = _HANDOFF["imgs"].get((y.device.index, y.data_ptr(), ("x", y.numel()))) if _HANDOFF["armed"] else None
    if img is not None and img.rowmap is rowmap and img.fmt == mode and img.cols == cols and img.rows == rowmap.cap:
        return img
    return None


def y_image(y, rows, cols, mode, rowmap):
    """the compact image of an LSTM output for its own weight-gradient GEMM: the one a consumer's backward left (same pass), else made here"""
    img = _handoff_take_x(y, cols, mode, rowmap) if rowmap is not None else None
    return img if img is not None else shared_image(y, rows, cols, mode, rowmap)


def _handoff_take(t):
    return _HANDOFF["imgs"].pop((t.device.index, t.data_ptr(), tuple(t.shape)), None)


def _handoff_put_image_only(t, img):
    """as _handoff_put, for a gradient tensor whose fp32 storage was NOT written (only its image exists): a consumer that would
    read the fp32 values fails loudly (_require_written)"""
    _handoff_put(t, img)
    if not _HANDOFF["armed"]:
        raise RuntimeError("flowtron_amd: an image-only gradient outside a backward pass")
    _HANDOFF["unwritten"].add((t.device.index, t.data_ptr(), tuple(t.shape)))


def _require_written(t):
    if _HANDOFF["unwritten"] and (t.device.index, t.data_ptr(), tuple(t.shape)) in _HANDOFF["unwritten"]:
        raise RuntimeError("flowtron_amd: this gradient exists only as a 16-bit operand image (ft_lstm_persist_bwd_img); its consumer "
                           "must take the image (set FLOWTRON_LSTM_PERSIST_IMG=both to keep the fp32 copy)")


# Split-K weight-gradient GEMMs accumulate with fp32 atomics into an output that must start at zero.  One memset per output was 37
# dispatches per training step (hipMemset2DAsync / zeros_like: 0.28 ms of ~5 us fills, profiles/r04_final_step_timeline.txt);
# the outputs of a backward pass now come out of ONE zeroed allocation sized by the previous pass's demand, and the GEMMs run with
# beta = 1 ("C holds the addend").  (_ZSLAB_ON = False: one zeroed tensor per output.)
_ZSLAB = {"buf": None, "off": 0, "need": 0, "last": 0, "task": None}
_ZSLAB_ON = True


def zeroed(shape, device):
    """a zero-filled fp32 tensor for an accumulating kernel; inside a backward pass a view into the pass's zeroed slab"""
    n = 1
    for d in shape:
        n *= int(d)
    gt = _current_graph_task() if _ZSLAB_ON else -1
    if gt == -1:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    z = _ZSLAB
    if z["task"] != (gt, str(device)):
        z["last"], z["need"], z["off"], z["task"] = max(z["need"], 0), 0, 0, (gt, str(device))
        z["buf"] = torch.zeros(z["last"], device=device, dtype=torch.float32) if z["last"] > 0 else None
    step = (n + 63) // 64 * 64                           # 256-byte aligned slices (vector atomics / float4 epilogues)
    z["need"] += step
    if z["buf"] is not None and z["off"] + step <= z["buf"].numel():
        v = z["buf"][z["off"]:z["off"] + n].view(shape)
        z["off"] += step
        return v
    return torch.zeros(shape, device=device, dtype=torch.float32)


_ARENA_GRADS = True


def weight_grad_out(W):
    """the zero-filled fp32 output of W's weight-gradient kernel (split-K / atomic accumulation): inside a backward pass the slice of
    the flat gradient arena the parameter's .grad will live in anyway (dist.FlatArena.grad_view_for_pass: no copy into the arena
    afterwards), else a piece of the pass's zeroed slab.  (_ARENA_GRADS = False: always the slab.)"""
    if _ARENA_GRADS and _ZSLAB_ON:
        gt = _current_graph_task()
        if gt != -1:
            from .dist import arena_slot
            slot = arena_slot(W)
            if slot is not None:
                v = slot[0].grad_view_for_pass(slot[1], gt, W.shape)
                if v is not None:
                    return v
    return zeroed(W.shape, W.device)


# An activation with several consumers (h_att feeds the query projection AND the decoder LSTM's input projection; the encoder output
# feeds the key and value projections of every flow) gets one input gradient per consumer, which autograd then adds: a [T,B,1024]
# fp32 add of 340 MB of traffic per flow (49 us) for h_att.  The image-path Linear backward instead ACCUMULATES into the buffer the
# first consumer of the same tensor produced in this pass (its dX GEMM runs with beta = 1) and returns None for that input -- the
# engine hands the producer the one buffer.  Safe because the engine runs an activation's producer only after ALL its consumers,
# and because forward activations never share an address while alive.  Buffers are held weakly (a weak reference survives while
# the engine's input buffer owns the tensor and dies with it): nothing outlives its consumer, and a buffer the engine has folded
# into another tensor (a third, foreign contribution arriving first and adding ours in place) is simply not found again.
# Caveat, by construction outside this model: if a FOREIGN owner (a user hook that stores gradient tensors) keeps the first buffer
# alive AND a third contribution of another kind makes the engine replace it by an out-of-place sum, a later accumulation would
# go into the orphan -- h_att and the encoder output have image-path consumers only; _DX_INPLACE = False restores autograd's adds.
_DX_INPLACE = True
_DXACC = {"task": None, "bufs": {}}


def _dx_buffer(x):
    """(buffer for the input gradient of x, accumulate: the buffer already holds another consumer's contribution)"""
    gt = _current_graph_task() if _DX_INPLACE else -1
    if gt == -1:
        return torch.empty_like(x), False
    d = _DXACC
    if d["task"] != gt:
        d["task"], d["bufs"] = gt, {}
    key = (x.device.index, x.data_ptr(), tuple(x.shape))
    ref = d["bufs"].get(key)
    buf = ref() if ref is not None else None
    if buf is not None:
        return buf, True
    buf = torch.empty_like(x)
    d["bufs"][key] = _weakref.ref(buf)
    return buf, False


def colsum(x2d: torch.Tensor, rows: int, N: int, ld: int) -> torch.Tensor:
    out = torch.empty(N, device=x2d.device, dtype=torch.float32)
    L.check(L.lib().ft_colsum(L.ptr(x2d), L.ptr(out), rows, N, ld, L.stream()), "ft_colsum")
    return out


_LENS_CACHE = {}


def _lens_entry(lens: torch.Tensor):
    """int32 copy and element sum of a length vector, made ONCE per tensor (keyed by storage address + version): the same in_lens /
    out_lens reach the model, every flow, the three losses and their backward passes -- a copy + a reduction + a cast of four-byte
    kernels each time (~20 dispatches of ~5 us per training step)"""
    key = (lens.data_ptr(), lens._version, lens.numel(), lens.dtype, lens.device)
    e = _LENS_CACHE.get(key)
    if e is None or e[0]() is not lens:
        import weakref
        if len(_LENS_CACHE) > 64:
            _LENS_CACHE.clear()
        i32 = lens if (lens.dtype == torch.int32 and lens.is_contiguous()) else lens.to(dtype=torch.int32).contiguous()
        e = _LENS_CACHE[key] = (weakref.ref(lens), i32, {})
    return e


def lens32(lens: torch.Tensor) -> torch.Tensor:
    return _lens_entry(lens)[1]


def lens_total(lens: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """float32 device scalar sum(lens) * scale (the normalisers of the losses, flowtron.py:206-243)"""
    e = _lens_entry(lens)
    t = e[2].get(scale)
    if t is None:
        t = e[2][scale] = e[1].sum().to(torch.float32) * scale if scale != 1.0 else e[1].sum().to(torch.float32)
    return t


# --------------------------------------------------------------------------
# Linear over one or two row-blocks of the weight:  y = act(sum_i x_i W[:, off_i:off_i+K_i]^T + b)
# (nn.Linear / 1x1 Conv1d / LSTM input projection call sites, flowtron.py:568-571, :758, :767-768)
# --------------------------------------------------------------------------
def linear_uses_images(mode, rows, N, xs):
    """whether LinearFn runs (and differentiates) this projection through shared 16-bit operand images"""
    return all(images_apply(mode, rows, N, x.shape[-1]) for x in xs) and all((x.shape[-1] % 8 == 0) for x in xs[:-1])


class LinearFn(torch.autograd.Function):
    """rowmap (RowMap | None): the inputs are time-major [T,B,K] activations of a padded batch; in the 16-bit image path the
    GEMMs then run over the VALID rows only (the reference packs them, flowtron.py:689-694): compact images, output rows
    scattered back, weight gradients reduced over compact rows.  Rows of padded frames are NOT written unless `fill` asks:
      "y"  in fill: forward output rows of padded frames = the utterance's first padded frame (which the GEMM computes: every
                    padded frame of an utterance has the same inputs), for consumers that walk all T frames;
      "dx" in fill: input-gradient rows of padded frames = 0, for consumers that reduce over all T frames.
    Call sites whose consumers only ever touch valid frames (the recurrences, the next compact GEMM) pass fill = ""."""

    @staticmethod
    def forward(ctx, W, bias, act, mode, rowmap, fill, *xs):
        xs = [_c(x) for x in xs]
        L.require_cuda(W, *xs)
        W = _c(W)
        N, Ktot = W.shape
        rows = xs[0].numel() // xs[0].shape[-1]
        c16 = "c16" in fill                       # y as 16-bit rows for a persistent recurrence (gx16_ok: implies the image path below)
        y = torch.empty(xs[0].shape[:-1] + (N,), device=W.device, dtype=op16_dtype(mode) if c16 else torch.float32)
        # mode = one operand format, or (forward, input gradient, weight gradient): mixed formats (the encoder convolutions keep
        # fp32 operands where 16-bit rounding is amplified, model.Encoder) take the per-GEMM path without shared images
        mode_dx = mode_dw = mode
        if isinstance(mode, tuple):
            mode, mode_dx, mode_dw = mode
        split_fwd = mode == "split3"                 # forward products from [hi | lo | hi] x [hi | hi | lo] images of the backward's format
        if split_fwd:
            assert len(xs) == 1 and act == L.ACT_NONE and L.is16(mode_dx) and mode_dx == mode_dw
            x2d = xs[0].reshape(rows, Ktot)
            split_imgs = None
            if Ktot % 32 == 0 and images_apply(mode_dx, rows, N, Ktot):
                lazy = _LAZY_COLS.pop(xs[0].data_ptr(), None)     # (an Im2colFn output whose values were never written: image from ITS source)
                xi = Bf16Image.split3_im2col(*lazy, mode_dx) if lazy is not None else Bf16Image.split3(x2d, mode_dx, False)
                wi = Bf16Image.of_weight(W, mode_dx, split=True)
                # K = 3 Ktot over only (rows / 128) x (N / 128) output tiles (160 for the encoder: a sixth of the chip's workgroup slots,
                # 240 k-steps each: 130 us): split-K fills the chip -- in its deterministic form (see _ENC_SPLITK)
                gemm_img(xi, 0, xi.ptr(), wi, 0, wi.ptr(), y, rows, N, 3 * Ktot, N, bias=bias, splitk=_ENC_SPLITK)
                split_imgs = (wi.view_cols(Ktot), [xi.view_cols(Ktot)])      # their [hi] blocks serve the backward's GEMMs as they are
            else:
                _materialize_col(xs[0])
                gemm_raw(x2d, W, y, rows, N, Ktot, Ktot, 1, 1, Ktot, N, bias=bias, mode=L.FT_F32)
            mode = L.FT_F32                          # (what the rest of this node records as its forward format)
        mixed = not (mode == mode_dx == mode_dw)
        ctx.mode_dx, ctx.mode_dw = mode_dx, mode_dw
        use_img = (not mixed) and linear_uses_images(mode, rows, N, xs)
        # fp32 forward, 16-bit backward: the backward makes the operand images itself (the forward had none to share)
        ctx.lazy_img_mode = mode_dx if (mixed and mode_dx == mode_dw and linear_uses_images(mode_dx, rows, N, xs)) else None
        if rowmap is not None and (not use_img or rows != rowmap.T * rowmap.B):
            rowmap = None
        ctx.imgs = split_imgs if split_fwd else None
        ctx.cat = False
        if use_img:
            w_img = Bf16Image.of_weight(W, mode)
            if len(xs) > 1 and rowmap is not None and _CAT_IMAGES:
                # [x_0 | x_1] as ONE compact image: one K loop over K_0 + K_1 (the 256 x 256 x 64 kernel applies at K >= 1536), one
                # weight-gradient GEMM -- instead of two K pieces accumulating through C (measured 259 + 316 us for the decoder
                # LSTM's input projection against ~330 for the single loop)
                x_cat = Bf16Image.cat_rows([x.reshape(rows, x.shape[-1]) for x in xs], mode, rowmap)
                ctx.imgs, ctx.cat = (w_img, [x_cat]), True
                gemm_img(x_cat, 0, x_cat.ptr(), w_img, 0, w_img.ptr(), y, rowmap.cap, N, Ktot, N, bias=bias, act=act, rowmap=rowmap, compact=1, c16=c16)
            else:
                x_imgs = [shared_image(x, rows, x.shape[-1], mode, rowmap) for x in xs]
                ctx.imgs = (w_img, x_imgs)          # reused by backward (dX reads W k-major, dW reads x k-major)
        off = 0
        for i, x in enumerate(xs):
            if ctx.cat or split_fwd:
                off = Ktot
                break
            K = x.shape[-1]
            last = i == len(xs) - 1
            if use_img:
                assert not c16 or len(xs) == 1
                gemm_img(x_imgs[i], 0, x_imgs[i].ptr(), w_img, 0, w_img.ptr(0, off), y, rows if rowmap is None else rowmap.cap, N, K, N,
                         bias=bias if last else None, act=act if last else L.ACT_NONE, beta=0.0 if i == 0 else 1.0,
                         rowmap=rowmap, compact=1, c16=c16)
            else:
                assert not c16, "16-bit output rows come from the image GEMMs only"
                gemm_raw(x, W[:, off:], y, rows, N, K, K, 1, 1, Ktot, N,
                         bias=bias if last else None, act=act if last else L.ACT_NONE,
                         beta=0.0 if i == 0 else 1.0, mode=mode)
            off += K
        assert off == Ktot
        if rowmap is not None and "y" in fill:
            rowmap.fill(y.reshape(rows, N), N, copy_separator=True)
        ctx.save_for_backward(W, y if act != L.ACT_NONE else None, *xs)
        ctx.act, ctx.mode, ctx.has_bias, ctx.rowmap, ctx.fill = act, mode, bias is not None, rowmap, fill
        return y

    @staticmethod
    def backward(ctx, dy):
        W, y, *xs = ctx.saved_tensors
        N, Ktot = W.shape
        rows = dy.numel() // N
        rowmap = ctx.rowmap
        if ctx.imgs is None and ctx.lazy_img_mode is not None:
            lm = ctx.lazy_img_mode
            ctx.imgs = (Bf16Image(W, mode=lm), [shared_image(x, rows, x.shape[-1], lm, rowmap) for x in xs])
        # an image handed over by the producer of dy (the LSTM backward) is looked up on dy AS IT ARRIVES: an image-only gradient is
        # a zero-stride NaN tensor (image_only_gradient) that must not be materialised
        d_img_in = _handoff_take(dy) if (ctx.act == L.ACT_NONE and ctx.imgs is not None) else None
        if d_img_in is not None and (d_img_in.fmt != ctx.imgs[0].fmt or d_img_in.rowmap is not rowmap
                                     or (ctx.has_bias and ctx.needs_input_grad[1] and d_img_in.colsum is None)):
            d_img_in = None
        if d_img_in is None:
            _require_written(dy)
            dy = _c(dy.float())
        # an activated layer over a row map on the image path: dpre = dy act'(pre) is formed INSIDE the conversion pass (image + bias
        # column sums, ft_bf16_image_rows_act_bwd) -- no fp32 dpre tensor
        fuse_act = ctx.act != L.ACT_NONE and ctx.imgs is not None and rowmap is not None and _FUSE_ACT_BWD
        if ctx.act != L.ACT_NONE and not fuse_act:
            dpre = torch.empty_like(dy)
            L.check(L.lib().ft_act_bwd(L.ptr(y), L.ptr(dy), L.ptr(dpre), dy.numel(), ctx.act, L.stream()), "ft_act_bwd")
        else:
            dpre = dy
        dW = None
        if ctx.needs_input_grad[0]:
            dW = weight_grad_out(W) if ctx.imgs is not None else torch.empty_like(W)      # (image path: split-K with beta = 1)
        want_db = ctx.has_bias and ctx.needs_input_grad[1]
        db = None
        dxs = []
        off = 0
        imgs = ctx.imgs
        if imgs is not None:
            w_img, x_imgs = imgs
            d_img = d_img_in                                                   # e.g. the LSTM backward already made it
            if d_img is None and fuse_act:
                d_img = Bf16Image.empty_rows(N, rowmap, w_img.fmt, dy.device)
                L.check(L.op16("ft_bf16_image_rows_act_bwd_acc", w_img.fmt)(L.ptr(dy), N, L.ptr(y), N, ctx.act, d_img.rows, N, L.ptr(d_img.buf),
                                                                             L.ptr(d_img.colsum), L.ptr(rowmap.map), L.ptr(rowmap.rows), L.stream()),
                        "ft_bf16_image_rows_act_bwd_acc")            # (empty_rows' colsum is a zeroed slab slice)
            elif d_img is None:
                d_img = Bf16Image(dpre.reshape(rows, N), colsum=want_db, mode=w_img.fmt, rowmap=rowmap)   # bias gradient rides on the conversion pass
            db = d_img.colsum if want_db else None
        if want_db and db is None:
            db = colsum(dpre, rows, N, N)
        mrows = rows if rowmap is None else rowmap.cap
        for i, x in enumerate(xs):
            K = x.shape[-1]
            if ctx.needs_input_grad[6 + i]:
                # dx[r,k] = sum_n dpre[r,n] W[n, off+k]
                if imgs is not None:
                    dx, acc = _dx_buffer(x)            # (another consumer of x may have left its contribution there: beta = 1)
                    gemm_img(d_img, 0, d_img.ptr(), w_img, 1, w_img.ptr(0, off), dx, mrows, K, N, K, beta=1.0 if acc else 0.0,
                             rowmap=rowmap, compact=1)
                    if rowmap is not None and "dx" in ctx.fill:
                        rowmap.fill(dx.reshape(rows, K), K, copy_separator=False)
                    if acc:
                        dx = None                      # autograd already holds the buffer through the first consumer
                else:
                    dx = torch.empty_like(x)
                    gemm_raw(dpre, W[:, off:], dx, rows, K, N, N, 1, Ktot, 1, K, mode=ctx.mode_dx)
                dxs.append(dx)
            else:
                dxs.append(None)
            if dW is not None and ctx.cat:
                if i == 0:      # one split-K GEMM over the concatenated image: dW[n, :] = sum_r dpre[r,n] [x_0 | x_1][r, :]
                    gemm_img(d_img, 1, d_img.ptr(), x_imgs[0], 1, x_imgs[0].ptr(), dW, N, Ktot, mrows, Ktot, beta=1.0, splitk=True, rowmap=rowmap, compact=2)
            elif dW is not None:
                # dW[n, off+k] = sum_r dpre[r,n] x[r,k]
                if imgs is not None:
                    gemm_img(d_img, 1, d_img.ptr(), x_imgs[i], 1, x_imgs[i].ptr(), dW[:, off:], N, K, mrows, Ktot, beta=1.0, splitk=True,
                             rowmap=rowmap, compact=2)
                    if rowmap is not None:
                        _handoff_put_x(x, x_imgs[i])
                else:
                    gemm_raw(dpre, x, dW[:, off:], N, K, rows, 1, N, K, 1, Ktot, mode=ctx.mode_dw, splitk=True)
            off += K
        ctx.imgs = None
        return (dW, db, None, None, None, None, *dxs)


_GATE_ON_CAT = True        # the gate layer on the decoder input projection's concatenated image (round 4)


def linear_gate_fusable(mode, rowmap, xs, N):
    """whether LinearGateFn applies: two inputs over a row map on the concatenated-image path (LinearFn's own conditions)"""
    if not (_GATE_ON_CAT and _CAT_IMAGES and rowmap is not None and len(xs) == 2):
        return False
    rows = xs[0].numel() // xs[0].shape[-1]
    return rows == rowmap.T * rowmap.B and linear_uses_images(mode, rows, N, xs)


class LinearGateFn(torch.autograd.Function):
    """y = [x0 | x1] W^T + b  AND  gate = [x0 | x1] wg^T + bg  from ONE concatenated compact image (flowtron.py:758-761: the decoder
    LSTM's input and the gate layer both read [h_att ; ctx]).  The N = 1 gate projection is a GEMV over the image the GEMM has just
    used (ft_img_gemv_rows) instead of two fp32 GEMMs with one output column; in backward its input gradient dgate (x) wg rides on
    the projection's dX GEMMs as a rank-1 epilogue term (no [T,B,1664] fp32 tensor, no accumulation pass) and its weight gradient
    is a GEMV^T over the same image.  Rows of padded frames: y as LinearFn (`fill`), gate = the utterance's separator value."""

    @staticmethod
    def forward(ctx, W, bias, Wg, bg, mode, rowmap, fill, x0, x1):
        xs = [_c(x0), _c(x1)]
        L.require_cuda(W, Wg, *xs)
        W, Wg = _c(W), _c(Wg)
        N, Ktot = W.shape
        assert Wg.shape == (1, Ktot) and xs[0].shape[-1] + xs[1].shape[-1] == Ktot
        rows = rowmap.T * rowmap.B
        c16 = "c16" in fill                       # y as 16-bit rows for a persistent recurrence (gx16_ok)
        y = torch.empty(xs[0].shape[:-1] + (N,), device=W.device, dtype=op16_dtype(mode) if c16 else torch.float32)
        gate = torch.empty(xs[0].shape[:-1] + (1,), device=W.device, dtype=torch.float32)
        w_img = Bf16Image.of_weight(W, mode)
        x_cat = Bf16Image.cat_rows([x.reshape(rows, x.shape[-1]) for x in xs], mode, rowmap)
        gemm_img(x_cat, 0, x_cat.ptr(), w_img, 0, w_img.ptr(), y, rowmap.cap, N, Ktot, N, bias=bias, rowmap=rowmap, compact=1, c16=c16)
        L.check(L.op16("ft_img_gemv_rows", mode)(L.ptr(x_cat.buf), x_cat.ld, Ktot, L.ptr(Wg), L.ptr(bg), L.ptr(gate), 1, L.ptr(rowmap.map),
                                                 L.ptr(rowmap.rows), L.ptr(rowmap.lens), rowmap.T, rowmap.B, L.stream()), "ft_img_gemv_rows")
        if "y" in fill:
            rowmap.fill(y.reshape(rows, N), N, copy_separator=True)
        ctx.save_for_backward(W, Wg, *xs)
        ctx.imgs = (w_img, x_cat)
        ctx.mode, ctx.has_bias, ctx.has_gbias, ctx.rowmap, ctx.fill = mode, bias is not None, bg is not None, rowmap, fill
        return y, gate

    @staticmethod
    def backward(ctx, dy, dgate):
        W, Wg, *xs = ctx.saved_tensors
        N, Ktot = W.shape
        rowmap = ctx.rowmap
        rows = rowmap.T * rowmap.B
        w_img, x_cat = ctx.imgs
        want_db = ctx.has_bias and ctx.needs_input_grad[1]
        d_img = _handoff_take(dy)                      # the LSTM backward's dgates image (possibly the ONLY form dy exists in)
        if d_img is not None and (d_img.fmt != w_img.fmt or d_img.rowmap is not rowmap or (want_db and d_img.colsum is None)):
            d_img = None
        if d_img is None:
            _require_written(dy)
            d_img = Bf16Image(_c(dy.float()).reshape(rows, N), colsum=want_db, mode=w_img.fmt, rowmap=rowmap)
        db = d_img.colsum if want_db else None
        rank1 = None
        if dgate is not None:
            dgate = _c(dgate)
            _require_written(dgate)
        dxs, off = [], 0
        for i, x in enumerate(xs):
            K = x.shape[-1]
            if ctx.needs_input_grad[7 + i]:
                dx, acc = _dx_buffer(x)
                if dgate is not None:
                    rank1 = (dgate, Wg.data_ptr() + 4 * off)
                gemm_img(d_img, 0, d_img.ptr(), w_img, 1, w_img.ptr(0, off), dx, rowmap.cap, K, N, K, beta=1.0 if acc else 0.0,
                         rowmap=rowmap, compact=1, rank1=rank1)
                if "dx" in ctx.fill:
                    rowmap.fill(dx.reshape(rows, K), K, copy_separator=False)
                dxs.append(None if acc else dx)
            else:
                dxs.append(None)
            off += K
        dW = None
        if ctx.needs_input_grad[0]:
            dW = weight_grad_out(W)
            gemm_img(d_img, 1, d_img.ptr(), x_cat, 1, x_cat.ptr(), dW, N, Ktot, rowmap.cap, Ktot, beta=1.0, splitk=True, rowmap=rowmap, compact=2)
        dWg = dbg = None
        if dgate is not None and (ctx.needs_input_grad[2] or (ctx.has_gbias and ctx.needs_input_grad[3])):
            acc = zeroed((Ktot + 1,), W.device)
            L.check(L.op16("ft_img_gemv_rows_bwd", w_img.fmt)(L.ptr(x_cat.buf), x_cat.ld, Ktot, L.ptr(dgate), 1, L.ptr(acc), acc.data_ptr() + 4 * Ktot,
                                                              L.ptr(rowmap.map), L.ptr(rowmap.rows), rowmap.cap, L.stream()), "ft_img_gemv_rows_bwd")
            dWg = acc[:Ktot].reshape(1, Ktot) if ctx.needs_input_grad[2] else None
            dbg = acc[Ktot:] if (ctx.has_gbias and ctx.needs_input_grad[3]) else None
        elif dgate is None:
            dWg = torch.zeros_like(Wg) if ctx.needs_input_grad[2] else None
            dbg = torch.zeros(1, device=W.device, dtype=torch.float32) if (ctx.has_gbias and ctx.needs_input_grad[3]) else None
        ctx.imgs = None
        return (dW, db, dWg, dbg, None, None, None, *dxs)


def linear(xs, W, bias=None, act=L.ACT_NONE, mode=None, rowmap=None, fill="y+dx"):
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    return LinearFn.apply(W, bias, act, L.mfma_mode() if mode is None else mode, rowmap, fill, *xs)


# --------------------------------------------------------------------------
# embedding gather (flowtron.py:873-874)
# --------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, W, run_stride=0):
        L.require_cuda(ids, W)
        ids = _c(ids.reshape(-1).to(torch.int64))
        W = _c(W)
        out = torch.empty(ids.numel(), W.shape[1], device=W.device, dtype=torch.float32)
        L.check(L.lib().ft_embedding_fwd(L.ptr(ids), L.ptr(W), L.ptr(out), ids.numel(), W.shape[1], W.shape[1], L.stream()),
                "ft_embedding_fwd")
        ctx.save_for_backward(ids, W)
        ctx.run_stride = int(run_stride)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, W = ctx.saved_tensors
        dout = _c(dout)
        dW = weight_grad_out(W)                     # zero-filled; the scatter-add accumulates
        if ctx.run_stride > 0:
            L.check(L.lib().ft_embedding_bwd_runs(L.ptr(ids), L.ptr(dout), L.ptr(dW), ids.numel(), W.shape[1], W.shape[1], ctx.run_stride,
                                                  L.stream()), "ft_embedding_bwd_runs")
        else:
            L.check(L.lib().ft_embedding_bwd(L.ptr(ids), L.ptr(dout), L.ptr(dW), ids.numel(), W.shape[1], W.shape[1], L.stream()),
                    "ft_embedding_bwd")
        return None, dW, None


def embedding(ids, W, run_stride=0):
    """run_stride > 0: the ids repeat with this period (row r carries ids[r % run_stride]'s value in practice) -- the backward then
    sums runs in registers (ft_embedding_bwd_runs); any ids are handled correctly"""
    return EmbeddingFn.apply(ids, W, run_stride)


# --------------------------------------------------------------------------
# encoder conv as im2col + GEMM, masked instance norm + relu (+dropout mask)
# --------------------------------------------------------------------------
# Im2colFn outputs whose VALUES have not been written (lazy = True): data_ptr -> (x, lens, KW).  conv_norm_relu passes such a matrix
# straight to the split-image Linear, whose forward makes the operand image from x itself (Bf16Image.split3_im2col) and whose backward
# reads only that image; any other reader must call _materialize_col first.
_LAZY_COLS = {}


def _materialize_col(col):
    src = _LAZY_COLS.pop(col.data_ptr(), None)
    if src is not None:
        x, lens, KW = src
        Lx, B, Cc = x.shape
        L.check(L.lib().ft_im2col(L.ptr(x), L.ptr(col), L.ptr(lens), Lx, B, Cc, KW, L.stream()), "ft_im2col")


class Im2colFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lens, KW, lazy=False):
        x = _c(x)
        L.require_cuda(x, lens)
        Lx, B, Cc = x.shape
        col = torch.empty(Lx, B, Cc * KW, device=x.device, dtype=torch.float32)
        if lazy:
            _LAZY_COLS[col.data_ptr()] = (x, lens, KW)
        else:
            L.check(L.lib().ft_im2col(L.ptr(x), L.ptr(col), L.ptr(lens), Lx, B, Cc, KW, L.stream()), "ft_im2col")
        ctx.save_for_backward(lens)
        ctx.dims = (Lx, B, Cc, KW)
        return col

    @staticmethod
    def backward(ctx, dcol):
        (lens,) = ctx.saved_tensors
        Lx, B, Cc, KW = ctx.dims
        dcol = _c(dcol)
        dx = torch.empty(Lx, B, Cc, device=dcol.device, dtype=torch.float32)
        L.check(L.lib().ft_col2im(L.ptr(dcol), L.ptr(dx), L.ptr(lens), Lx, B, Cc, KW, L.stream()), "ft_col2im")
        return dx, None, None, None


class InstNormReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, keep, lens, eps):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        L.require_cuda(x, gamma, beta, keep, lens)
        if keep is not None:
            keep = _c(keep)
        Lx, B, Cc = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        L.check(L.lib().ft_instnorm_relu_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(keep), L.ptr(lens), L.ptr(y),
                                             L.ptr(mean), L.ptr(rstd), Lx, B, Cc, eps, L.stream()), "ft_instnorm_relu_fwd")
        ctx.save_for_backward(x, y, gamma, keep, lens, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, keep, lens, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        Lx, B, Cc = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty_like(gamma)
        db = torch.empty_like(gamma)
        L.check(L.lib().ft_instnorm_relu_bwd(L.ptr(x), L.ptr(y), L.ptr(dy), L.ptr(gamma), L.ptr(keep), L.ptr(lens),
                                             L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dg), L.ptr(db), Lx, B, Cc, L.stream()),
                "ft_instnorm_relu_bwd")
        return dx, dg, db, None, None, None


def conv_norm_relu(x, lens, conv_w, conv_b, gamma, beta, keep=None, eps=1e-5, mode=None):
    """x [L,B,C] -> relu(masked_instance_norm(conv1d_k5(x))) * keep  (flowtron.py:499-502)."""
    Cout, Cin, KW = conv_w.shape
    # split-image forward (model.Encoder's default in the 16-bit modes): the column matrix is needed as an operand image only -- made from
    # x itself; the fp32 matrix stays unwritten (Im2colFn lazy) unless the Linear cannot take the image path after all
    rows, K = x.shape[0] * x.shape[1], Cin * KW
    lazy = (isinstance(mode, tuple) and mode[0] == "split3" and L.is16(mode[1]) and mode[1] == mode[2] and K % 32 == 0
            and images_apply(mode[1], rows, Cout, K))
    col = Im2colFn.apply(x, lens, KW, lazy)
    y = linear(col, conv_w.reshape(Cout, Cin * KW), conv_b, mode=mode)
    if lazy:
        _materialize_col(col)                    # (a no-op when the Linear took the image from x: it popped the entry)
    return InstNormReluFn.apply(y, gamma, beta, keep, lens, eps)


# --------------------------------------------------------------------------
# length-masked LSTM sequence (packed nn.LSTM semantics, flowtron.py:689-694, :505-512)
# --------------------------------------------------------------------------
# --------------------------------------------------------------------------
# persistent recurrence kernels (csrc/lstm_persist.hip): one launch per sequence, W_hh resident in registers
# --------------------------------------------------------------------------
# The kernels raise a device status word instead of hanging when their hand-off waits time out (grid not co-resident,
# partitioned device, a foreign kernel holding CUs).  Nothing here stalls the stream and nothing kills a run:
#   * a self-test on first use (synchronous, once per device) decides whether the path is usable at all;
#   * afterwards every launch first looks at an asynchronous host copy of the word made after the PREVIOUS launch: a failure
#     surfaces one launch late, is logged ONCE, and switches this device to the launch-per-step kernels for good;
#   * the step that contained the failed launch never reaches the weights: the word stays set until the optimizer (or the DP
#     wrapper, before the gradient all-reduce) has enqueued `if (status) grad[0] = NaN` (ft_poison_if_nonzero), and the fused
#     RAdam kernel drops an update whose global gradient norm is not finite (ft_radam_step) -- on every rank, without a host
#     synchronisation.  While the word is set, further persistent launches of that step abort at their first wait.
_PERSIST = {}


class _PersistState:
    def __init__(self, device):
        self.status = torch.zeros(1, device=device, dtype=torch.int32)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None
        self.usable = None          # None = not tested yet
        self.fail_seen = False      # the host has noticed a failure whose status word has not been consumed (poisoned) yet
        self.failures = 0


def _persist_state(device):
    st = _PERSIST.get(device)
    if st is None:
        st = _PERSIST[device] = _PersistState(device)
    return st


def persist_status(device):
    return _persist_state(device).status


def _persist_failed(st, device, code, where):
    import warnings
    st.usable = False
    st.fail_seen = True
    st.failures += 1
    st.event = None
    warnings.warn("flowtron_amd: a persistent recurrence launch on %s did not complete (status %d, seen %s): the 256-workgroup grid was "
                  "not co-resident / the XCD census failed.  That optimizer step is dropped (non-finite-norm guard) and this device "
                  "uses the launch-per-step kernels from now on." % (device, code, where))


def _persist_watch(device):
    """look at the asynchronous host copy of the status word made after the previous persistent launch (no stall)"""
    st = _persist_state(device)
    if st.event is not None and st.event.query():
        code = int(st.host[0])
        st.event = None
        if code != 0 and not st.fail_seen:
            _persist_failed(st, device, code, "one launch late")
    return st


def persist_consume_failure(device):
    """Called by whoever has just enqueued the poison of this step's gradients (optim.poison_from_status): once the host has seen
    the failure, the status word is cleared BEHIND the poison kernel on the stream, so exactly the affected steps are dropped."""
    st = _PERSIST.get(device)
    if st is not None and st.fail_seen:
        st.status.zero_()
        st.fail_seen = False


PERSIST_LAUNCHES = 0          # persistent (whole-chip, co-resident) kernel launches so far: dist.py keeps collectives away from them


BILSTM_PERSIST_LAUNCHES = 0   # launches of the encoder's persistent bidirectional kernels (csrc/bilstm_persist.hip)


def _persist_arm(st, bilstm=False):
    global PERSIST_LAUNCHES, BILSTM_PERSIST_LAUNCHES
    if bilstm:
        BILSTM_PERSIST_LAUNCHES += 1
    else:
        PERSIST_LAUNCHES += 1
    if st.event is not None:
        # the previous asynchronous copy has not been looked at yet (the host runs about a step ahead of the device): one copy in
        # flight is enough -- the word stays set until it is consumed, so the next copy sees a failure of any launch before it --
        # and a D2H copy per launch costs the stream ~10 us each (copy dispatch + the gap behind it; profiles/r04_v1_step_timeline.txt)
        return
    st.host.copy_(st.status, non_blocking=True)
    st.event = torch.cuda.Event()
    st.event.record()


def check_persist_status(raise_on_failure=True):
    """Host check (one sync) of every persistent-kernel status word.  Tests / bench / inference call it where a garbage result
    must not go unnoticed; returns True when everything completed.  On a failure the device is switched to the launch-per-step
    kernels; with raise_on_failure the caller gets a RuntimeError (the result it just computed is invalid), otherwise False
    (the caller re-runs, e.g. AR_Step.infer on the staged decode chain)."""
    ok = True
    for dev, st in _PERSIST.items():
        code = int(st.status.item())
        if code != 0:
            ok = False
            if not st.fail_seen:
                _persist_failed(st, dev, code, "by a synchronous check")
            st.status.zero_()
            st.fail_seen = False
            if raise_on_failure:
                raise RuntimeError("persistent kernel timed out on %s (status %d): the 256-workgroup grid was not co-resident; the "
                                   "launch-per-step kernels are used from now on (FLOWTRON_LSTM_PERSIST=0 selects them up front)" % (dev, code))
    return ok


PERSIST_BWD_CODE = 21        # ft_lstm_persist_bwd's one transport: the reduce-scatter kernel (csrc/lstm_persist.hip)


def _persist_selftest(device, ng=1):
    """tiny forward + backward through the persistent kernels the TRAINING step launches -- the roles forward at 4 and at 8 rows per
    XCD group with two roles (csrc/lstm_roles.hip: its own launch context and LDS sizes), the reduce-scatter backward once through
    ft_lstm_persist_bwd and once through the image-only entry the step itself uses (ft_lstm_persist_bwd_img) -- checked
    synchronously, once per device (a device where only some of the kernels are co-resident must not pass the self-test
    and then drop its first optimizer steps)."""
    H, B, T = 1024, 8, 3
    f = dict(device=device, dtype=torch.float32)
    gx, w = torch.zeros(T, B, 4 * H, **f), torch.zeros(4 * H, H, **f)
    lens = torch.full((B,), T, dtype=torch.int32, device=device)
    y, gates, cell = torch.empty(T, B, H, **f), torch.empty(T, B, 4 * H, **f), torch.empty(T, B, H, **f)
    dgx = torch.empty(T, B, 4 * H, **f)
    work = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=device, dtype=torch.uint8)
    st = _persist_state(device)
    st.usable = True                     # (roles_launch consults the state it is about to establish)
    try:
        wimg = roles_wimg(w, L.FT_BF16, False)
        roles_launch([fwd_role(gx, lens, y, gates, cell, wimg)], 4, L.FT_BF16, device)
        roles_launch([fwd_role(gx, lens, y, gates, cell, wimg), fwd_role(gx, lens, y, gates, cell, wimg)], 8, L.FT_BF16, device)
        L.check(L.lib().ft_lstm_persist_bwd(L.ptr(y), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx), L.ptr(work),
                                            L.ptr(st.status), T, B, H, PERSIST_BWD_CODE, L.stream()), "ft_lstm_persist_bwd")
        if _PERSIST_IMG != "0":
            rows, ld = T * B + B, (4 * H + 255) // 256 * 256
            dimg = torch.empty(L.lib().ft_bf16_image_bytes(rows, 4 * H), device=device, dtype=torch.uint8)
            dbias = torch.zeros(4 * H, **f)
            L.check(L.lib().ft_lstm_persist_bwd_img(L.ptr(y), H, L.ptr(w), L.ptr(lens), L.ptr(gates), L.ptr(cell), None, L.ptr(work),
                                                    L.ptr(st.status), T, B, H, PERSIST_BWD_CODE, L.ptr(dimg), ld,
                                                    dimg.numel() // (2 * ld), L.ptr(dbias), L.stream()), "ft_lstm_persist_bwd_img")
        ok = int(st.status.item()) == 0
    except RuntimeError:
        ok = False
    st.status.zero_()
    st.event = None
    if not ok:
        import warnings
        warnings.warn("flowtron_amd: the persistent LSTM kernels are not usable on %s (grid not co-resident / XCD census failed); "
                      "using the launch-per-step kernels" % (device,))
    return ok


def persist_usable(device):
    """True if the persistent kernels passed their self-test on this device (256 co-resident workgroups, XCD census)."""
    if not L.lib().ft_lstm_persist_supported(8, 1024):
        return False
    st = _persist_state(torch.device(device))
    if st.usable is None:
        st.usable = _persist_selftest(torch.device(device), int(_os.environ.get("FLOWTRON_LSTM_PERSIST", "1")) or 1)
    return bool(st.usable)


def lstm_persist_groups(B, H, reverse, mode, device=None):
    """1 when the persistent recurrence kernels take this shape in one launch per sequence (H 1024, B <= 32, forward direction,
    16-bit operands, a device that passed the self-test), else 0 = the launch-per-step kernels.  FLOWTRON_LSTM_PERSIST=0 switches
    the persistent kernels off (the yardstick path of the tests)."""
    ng = int(_os.environ.get("FLOWTRON_LSTM_PERSIST", "1"))
    if ng not in (0, 1):
        raise ValueError("FLOWTRON_LSTM_PERSIST must be 0 or 1 (the transport variants were removed in round 6)")
    if not ng or reverse or not L.is16(mode) or not L.lib().ft_lstm_persist_supported(B, H):
        return 0
    if device is not None:
        st = _persist_state(device)
        if st.usable is None:
            st.usable = _persist_selftest(device, ng)
        if not st.usable:
            return 0
    return 1


def lstm_persist_slices(B, H, reverse, mode, device=None):
    """True for a batch wider than the 4-row kernels hold (B > 32) where the roles kernels apply: 8 rows per XCD group up to B 64, 16
    up to 128 per launch forward / slices of 64 backward (roles_plan; 1.93 us per step at B 64 against 2 x 1.76 for two sliced launches
    of the round-5 kernel, 6.8 us for the launch-per-step kernels at B 48).  False: one launch suffices, or the shape is not theirs."""
    if B <= 32:
        return False
    return bool(lstm_persist_groups(32, H, reverse, mode, device))


# b_ih + b_hh of the LSTMs of a flow (and of the encoder) in ONE launch each per forward (a multi-tensor add; eight elementwise launches
# of ~5 us stood there: VERDICT r5 weak #12).  model.Flowtron.forward brackets its pass with bias_sums_begin / bias_sums_end; outside such a
# bracket -- and for a pair the bracket did not list -- bias_sum is the plain add.
_BIAS_SUMS = {}


class _BiasSumsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *flat):
        n = len(flat) // 2
        return tuple(torch._foreach_add(list(flat[:n]), list(flat[n:])))

    @staticmethod
    def backward(ctx, *gs):
        return tuple(gs) + tuple(gs)


def bias_sums_begin(groups):
    """groups: lists of (b_ih, b_hh) pairs; one launch (and one autograd node) per group, made at the group's FIRST request -- the
    node then sits where the first of the adds it replaces sat, and backward reaches it in the same order"""
    _BIAS_SUMS.clear()
    for pairs in groups:
        pairs = [(a, b) for a, b in pairs if a.is_cuda and b.is_cuda and a.shape == b.shape]
        if len(pairs) < 2:
            continue
        group = {"pairs": pairs, "outs": None}
        for a, b in pairs:
            _BIAS_SUMS[(id(a), id(b))] = group


def bias_sums_end():
    _BIAS_SUMS.clear()


def bias_sum(a, b):
    g = _BIAS_SUMS.get((id(a), id(b)))
    if g is None:
        return a + b
    if g["outs"] is None:
        g["outs"] = _BiasSumsFn.apply(*[x for x, _ in g["pairs"]], *[y for _, y in g["pairs"]])
    for (x, y), o in zip(g["pairs"], g["outs"]):
        if x is a and y is b:
            return o
    return a + b


PERSIST_H = 1024                  # the hidden size the persistent recurrence kernels are built for
_PAD_H = True    # hidden sizes below it on the persistent kernels through zero-padded gate blocks (tests set it False: the yardstick)


def lstm_pad_width(B, H, reverse, mode, device, T):
    """True where a layer with H < 1024 hidden units should run as its zero-padded 1024-unit twin (lstm_layer): the persistent kernels
    take the padded shape and the sequence is long enough for 1.3-1.6 us per step against the launch-per-step kernels' 5-6 to pay for the
    wider projection (4096 gate columns instead of 4 H)"""
    if not _PAD_H or H >= PERSIST_H or H < 256 or H % 8 or reverse or T < 64:      # (H < 256: > 4 x the gate columns and saved gates)
        return False
    return bool(lstm_persist_groups(min(B, 32), PERSIST_H, reverse, mode, device))


def pad_gate_blocks(w, pad_cols):
    """[4 H, X] (or [4 H]) -> [4 * 1024, X + pad_cols]: each of the four gate blocks zero-padded to 1024 rows, X to X + pad_cols columns"""
    H = w.shape[0] // 4
    if w.dim() == 1:
        return torch.nn.functional.pad(w.view(4, H), (0, PERSIST_H - H)).reshape(4 * PERSIST_H)
    return torch.nn.functional.pad(w.view(4, H, w.shape[1]), (0, pad_cols, 0, PERSIST_H - H)).reshape(4 * PERSIST_H, w.shape[1] + pad_cols)


def bilstm_persist_ok(B, H, mode, device):
    """the encoder-shaped persistent bidirectional kernels (H 256, B <= 32, 16-bit operands) on a device whose persistent grids
    passed the self-test; FLOWTRON_LSTM_PERSIST=0 keeps the launch-per-step pair chain"""
    # (the kernels form their groups from the XCD census like transports 1 / 9: not for the placement-independent fabric
    # transports 8 / 4 / 2, which are what one selects on a device where the census cannot come out)
    if int(_os.environ.get("FLOWTRON_LSTM_PERSIST", "1")) != 1:
        return False
    if not L.is16(mode) or not L.lib().ft_bilstm_persist_supported(B, H):
        return False
    st = _persist_state(device)
    if st.usable is None:
        st.usable = _persist_selftest(device, int(_os.environ.get("FLOWTRON_LSTM_PERSIST", "1")))
    return bool(st.usable)


# ---- round 6: rows per XCD group / time windows / two roles per launch (csrc/lstm_roles.hip) ------------------------------------
class RolesCtx:
    """Launch context of the ft_lstm_roles_* kernels on one device: hand-off sets + census counters (initialised once) and the
    launch counters of the two kernel kinds (a launch works in set phase & 1 and presets the other one for its successor)."""

    def __init__(self, device):
        self.buf = torch.empty(L.lib().ft_lstm_roles_ctx_bytes(), device=device, dtype=torch.uint8)
        L.check(L.lib().ft_lstm_roles_ctx_init(L.ptr(self.buf), L.stream()), "ft_lstm_roles_ctx_init")
        self.phase = [0, 0]          # forward, backward
        self.reset_rows = 8          # the largest R any later launch of a kind may use (roles_reset_rows)


_ROLES_CTX = {}


def roles_ctx(device):
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    c = _ROLES_CTX.get(device)
    if c is None:
        c = _ROLES_CTX[device] = RolesCtx(device)
    return c


def roles_wimg(w_hh, mode, backward):
    """MFMA fragment image of W_hh for the roles kernels (forward: gate-adjacent tiles; backward: the reduce-scatter layout)."""
    img = torch.empty(L.lib().ft_lstm_roles_wimg_bytes(w_hh.shape[1]), device=w_hh.device, dtype=torch.uint8)
    fn = L.op16("ft_lstm_roles_prepare_bwd" if backward else "ft_lstm_roles_prepare_fwd", mode)
    w = _c(w_hh)                 # (named: a temporary copy would be recycled by the allocator before the launch reads it)
    L.check(fn(L.ptr(w), L.ptr(img), w.shape[1], L.stream()), "ft_lstm_roles_prepare")
    return img


def fwd_role(gx, lens, y, gates, cell, wimg, t0=0, t1=None, state=None, b0=0, nb=None):
    """one forward recurrence of a roles launch over the batch rows [b0, b0 + nb) of time-major tensors, window [t0, t1)"""
    T, LB, H = gx.shape[0], gx.shape[1], y.shape[2]
    nb = LB - b0 if nb is None else nb
    return L.LstmFwdRole(gx.data_ptr() + gx.element_size() * 4 * H * b0, lens.data_ptr() + 4 * b0, y.data_ptr() + 4 * y.stride(1) * b0, y.stride(1),
                         gates.data_ptr() + 16 * H * b0 if gates is not None else None, cell.data_ptr() + 4 * H * b0 if cell is not None else None,
                         L.ptr(wimg), L.ptr(state[0]) if state is not None else None, L.ptr(state[1]) if state is not None else None,
                         nb, LB, t0, T if t1 is None else t1, int(gx.dtype != torch.float32))


def bwd_role(dy, lens, gates, cell, dgx, wimg, t0=0, t1=None, state=None, carry_in=False, dimg=None, b0=0, nb=None):
    T, LB, H = gates.shape[0], gates.shape[1], cell.shape[2]
    nb = LB - b0 if nb is None else nb
    return L.LstmBwdRole(dy.data_ptr() + 4 * dy.stride(1) * b0, dy.stride(1), lens.data_ptr() + 4 * b0, gates.data_ptr() + 16 * H * b0,
                         cell.data_ptr() + 4 * H * b0, dgx.data_ptr() + 16 * H * b0 if dgx is not None else None, L.ptr(wimg),
                         L.ptr(dimg.buf) if dimg is not None else None, dimg.ld if dimg is not None else 0,
                         dimg.buf.numel() // (2 * dimg.ld) if dimg is not None else 0, L.ptr(dimg.colsum) if dimg is not None else None,
                         L.ptr(state[0]) if state is not None else None, L.ptr(state[1]) if state is not None else None,
                         nb, LB, t0, T if t1 is None else t1, int(bool(carry_in)))


def roles_launch(roles, R, mode, device, backward=False, H=1024):
    """one launch of lstm_roles_fwd_k / _bwd_k: `roles` = one or two role structs; R rows per XCD group"""
    c = roles_ctx(device)
    if R > c.reset_rows:
        # wider groups than any launch on this context has preset for: start over with the sets preset for R rows (rare: the first
        # batch wider than 64 on a device)
        L.check(L.lib().ft_lstm_roles_ctx_init(L.ptr(c.buf), L.stream()), "ft_lstm_roles_ctx_init")
        c.phase, c.reset_rows = [0, 0], R
    st = _persist_watch(c.buf.device)
    k = 1 if backward else 0
    arr = ((L.LstmBwdRole if backward else L.LstmFwdRole) * len(roles))(*roles)
    fn = L.op16("ft_lstm_roles_bwd" if backward else "ft_lstm_roles_fwd", mode)
    L.check(fn(arr, len(roles), R, c.reset_rows, L.ptr(c.buf), c.phase[k], L.ptr(st.status), H, L.stream()), "ft_lstm_roles")
    c.phase[k] += 1
    _persist_arm(st)


def roles_plan(B, backward):
    """[(first row, rows, rows per XCD group)] of the roles launches that cover a batch of B rows"""
    cap = 64 if backward else 128
    out = []
    for b0 in range(0, B, cap):
        nb = min(cap, B - b0)
        out.append((b0, nb, 4 if nb <= 32 else 8 if nb <= 64 else 16))
    return out


_PAIR_CHUNKS_BWD = int(_os.environ.get("FLOWTRON_LSTM_PAIR_BWD", "0"))   # chunks of the pair's BACKWARD pipeline; 0 = sequential, -1 = as forward
_PAIR_CHUNKS = int(_os.environ.get("FLOWTRON_LSTM_PAIR", "6"))       # time chunks of the decoder layer pair pipeline; 0 = one recurrence per launch


def decoder_pair_chunks(B, H, mode, device, T):
    """number of time chunks the two decoder layers are pipelined over (DecoderPairFn), or 0: the pair needs both recurrences on
    R = 8 rows per XCD, four XCDs each (B <= 32), the persistent kernels' shape and a sequence long enough to chunk"""
    if _PAIR_CHUNKS <= 0 or B > 32 or T < 4 * _PAIR_CHUNKS or not lstm_persist_groups(B, H, False, mode, device):
        return 0
    return _PAIR_CHUNKS


def _chunk_edges(T, n):
    return [(k * T + n // 2) // n for k in range(n + 1)]


_EDGE_CACHE = {}


def _chunk_rowmaps(lens, edges, B):
    """per time chunk [e_k, e_k+1): the chunk as a padded batch of its own -- lengths clamp(len - e_k, 0, width) -- and its RowMap
    (the edge vectors are made on the device once per (edges, device): no host-to-device copy in a step)"""
    key = (tuple(edges), lens.device)
    ev = _EDGE_CACHE.get(key)
    if ev is None:
        T, n = edges[-1], len(edges) - 1
        k = torch.arange(n + 1, device=lens.device, dtype=torch.int64)
        e = torch.div(k * T + n // 2, n, rounding_mode="floor").to(torch.int32)
        if len(_EDGE_CACHE) > 64:
            _EDGE_CACHE.clear()                  # (drops the last-maps entries too)
        ev = _EDGE_CACHE[key] = (e[:-1].contiguous(), (e[1:] - e[:-1]).contiguous())
    # both flows of a forward pass chunk the same lengths the same way: the maps of the last (lens tensor, edges) are kept
    ck = (lens.data_ptr(), lens._version, key)
    if _EDGE_CACHE.get("last_key") == ck and _EDGE_CACHE["last_lens"]() is lens:
        return _EDGE_CACHE["last_maps"]
    lk = torch.minimum((lens[None, :] - ev[0][:, None]).clamp_(min=0), ev[1][:, None])
    maps = [RowMap(lk[k], edges[k + 1] - edges[k], B) for k in range(len(edges) - 1)]
    _EDGE_CACHE["last_key"], _EDGE_CACHE["last_lens"], _EDGE_CACHE["last_maps"] = ck, _weakref.ref(lens), maps
    return maps


class DecoderPairFn(torch.autograd.Function):
    """Both decoder LSTM layers of a flow (nn.LSTM(.., num_layers=2), flowtron.py:654-655, 689-694) as ONE pipeline over n time chunks:
    launch k runs layer 0 on chunk k (XCDs 0-3, 8 batch rows each) and layer 1 on chunk k - 1 (XCDs 4-7) CONCURRENTLY
    (csrc/lstm_roles.hip: two roles per launch, carried state), and layer 1's input projection of chunk k -- a chip-filling GEMM
    over the chunk's valid rows -- runs between launch k and launch k + 1.  Backward the other way round: layer 1 on chunk c, layer 0 on
    chunk c + 1, the chunk's dX GEMM (dy0 = dgates1 W_ih1) in between.  The dependency chain of the pair is (1 + 1/n) T steps of the
    8-row kernel instead of 2 T steps of the 4-row one; the arithmetic is that of two single launches (forward bit-identical)."""

    @staticmethod
    def forward(ctx, gx0, w_hh0, w_ih1, b_ih1, b_hh1, w_hh1, lens, mode, rowmap, gx_private, nchunks):
        gx0, w_hh0, w_ih1, w_hh1 = _c(gx0), _c(w_hh0), _c(w_ih1), _c(w_hh1)
        L.require_cuda(gx0, w_hh0, w_ih1, w_hh1, lens)
        T, B, H4 = gx0.shape
        H = H4 // 4
        dev = gx0.device
        f = dict(device=dev, dtype=torch.float32)
        y0, g0, c0 = torch.empty(T, B, H, **f), torch.empty(T, B, H4, **f), torch.empty(T, B, H, **f)
        y1, g1, c1 = torch.empty(T, B, H, **f), torch.empty(T, B, H4, **f), torch.empty(T, B, H, **f)
        g16 = gx0.dtype != torch.float32          # 16-bit gx rows (decoder_pair: gx16_ok): layer 1's chunk projections write them too
        gx1 = torch.empty(T, B, H4, device=dev, dtype=gx0.dtype)
        # (carried state: written by a window before the next one reads it -- a group that runs in window k ran in window k - 1)
        st0, st1 = torch.empty(2, B, H, **f), torch.empty(2, B, H, **f)
        wi0, wi1 = roles_wimg(w_hh0, mode, False), roles_wimg(w_hh1, mode, False)
        w_img = Bf16Image.of_weight(w_ih1, mode)
        b1 = bias_sum(b_ih1, b_hh1).detach()
        edges = _chunk_edges(T, nchunks)
        rms = _chunk_rowmaps(lens, edges, B)
        for k in range(nchunks + 1):
            roles = []
            if k < nchunks:
                roles.append(fwd_role(gx0, lens, y0, g0, c0, wi0, edges[k], edges[k + 1], st0))
            if k > 0:
                roles.append(fwd_role(gx1, lens, y1, g1, c1, wi1, edges[k - 1], edges[k], st1))
            roles_launch(roles, 8 if len(roles) == 2 else 4, mode, dev)
            if k < nchunks:
                # layer 1's input projection of chunk k over its valid rows, scattered into gx1[e_k : e_k+1]
                a, b = edges[k], edges[k + 1]
                x_img = Bf16Image(y0[a:b].reshape((b - a) * B, H), mode=mode, rowmap=rms[k])
                gemm_img(x_img, 0, x_img.ptr(), w_img, 0, w_img.ptr(), gx1[a:b], rms[k].cap, H4, H, H4, bias=b1, rowmap=rms[k], compact=1, c16=g16)
        ctx.save_for_backward(w_hh0, w_ih1, w_hh1, lens, y0, g0, c0, y1, g1, c1)
        ctx.mode, ctx.rowmap, ctx.gx_private, ctx.nchunks = mode, rowmap, bool(gx_private), nchunks
        ctx.chunk_maps, ctx.w_img, ctx.gx_dtype = (edges, rms), w_img, gx0.dtype
        return y1

    @staticmethod
    def backward(ctx, dy1):
        w_hh0, w_ih1, w_hh1, lens, y0, g0, c0, y1, g1, c1 = ctx.saved_tensors
        dy1 = _c(dy1)
        T, B, H = y1.shape
        H4, mode, rm, n = 4 * H, ctx.mode, ctx.rowmap, ctx.nchunks
        dev = dy1.device
        f = dict(device=dev, dtype=torch.float32)
        edges, rms = ctx.chunk_maps
        # layer 0's dgates leave as the compact 16-bit image alone where the only consumer is the input projection's backward
        img_ok = rm is not None and _PERSIST_IMG != "0" and images_apply(mode, H4, H, (T - 1) * B) and (ctx.gx_private or _PERSIST_IMG == "both")
        img_only = img_ok and _PERSIST_IMG != "both" and not torch.is_anomaly_enabled()
        w_img = ctx.w_img
        ctx.w_img = None
        nb = _PAIR_CHUNKS_BWD if _PAIR_CHUNKS_BWD >= 0 else n
        if nb == 0 and rm is not None and img_ok and ctx.needs_input_grad[1] and ctx.needs_input_grad[5]:
            return DecoderPairFn._backward_sequential(ctx, dy1, w_img, img_only)      # (before the pipeline's buffers: four fills less per pass)
        dgx1 = torch.empty(T, B, H4, **f)
        dy0 = torch.empty(T, B, H, **f)
        d_img0 = Bf16Image.empty_rows(H4, rm, mode, dev) if img_ok else None
        dgx0 = None if img_only else torch.empty(T, B, H4, **f)
        sb1 = (torch.zeros(B, H4, **f), torch.zeros(B, H, **f))
        sb0 = (torch.zeros(B, H4, **f), torch.zeros(B, H, **f))
        if nb != n and nb > 0:
            n = nb
            edges = _chunk_edges(T, n)
            rms = _chunk_rowmaps(lens, edges, B)
        wb0, wb1 = roles_wimg(w_hh0, mode, True), roles_wimg(w_hh1, mode, True)
        for j in range(n + 1):
            c = n - 1 - j                                        # layer 1's chunk in this launch; layer 0 runs chunk c + 1
            roles = []
            if j < n:
                roles.append(bwd_role(dy1, lens, g1, c1, dgx1, wb1, edges[c], edges[c + 1], sb1, carry_in=c < n - 1))
            if j > 0:
                roles.append(bwd_role(dy0, lens, g0, c0, dgx0, wb0, edges[c + 1], edges[c + 2], sb0, carry_in=c + 1 < n - 1, dimg=d_img0))
            roles_launch(roles, 8 if len(roles) == 2 else 4, mode, dev, backward=True)
            if j < n:
                # dy0 of chunk c = dgates1 W_ih1 over the chunk's valid rows
                a, b = edges[c], edges[c + 1]
                d_img = Bf16Image(dgx1[a:b].reshape((b - a) * B, H4), mode=mode, rowmap=rms[c])
                gemm_img(d_img, 0, d_img.ptr(), w_img, 1, w_img.ptr(), dy0[a:b], rms[c].cap, H, H4, H, rowmap=rms[c], compact=1)
        # weight / bias gradients over the whole sequence (compact rows; the one-step shift of dW_hh = one compact row)
        dW_hh0 = dW_ih1 = dW_hh1 = db1 = None
        need = ctx.needs_input_grad
        compact_ok = rm is not None and T > 1 and images_apply(mode, H4, H, (T - 1) * B)
        if need[1] or need[2] or need[3] or need[4] or need[5]:
            if compact_ok:
                d_img1 = Bf16Image(dgx1.reshape(T * B, H4), colsum=True, mode=mode, rowmap=rm)
                db1 = d_img1.colsum
                y0_img, y1_img = shared_image(y0, T * B, H, mode, rm), shared_image(y1, T * B, H, mode, rm)
                if need[5]:
                    dW_hh1 = weight_grad_out(w_hh1)
                    gemm_img(d_img1, 1, d_img1.ptr(1), y1_img, 1, y1_img.ptr(0), dW_hh1, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2, k_shift=1)
                if need[2]:
                    dW_ih1 = weight_grad_out(w_ih1)
                    gemm_img(d_img1, 1, d_img1.ptr(), y0_img, 1, y0_img.ptr(), dW_ih1, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2)
                if need[1]:
                    d0 = d_img0 if d_img0 is not None else Bf16Image(dgx0.reshape(T * B, H4), colsum=True, mode=mode, rowmap=rm)
                    dW_hh0 = weight_grad_out(w_hh0)
                    gemm_img(d0, 1, d0.ptr(1), y0_img, 1, y0_img.ptr(0), dW_hh0, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2, k_shift=1)
                    d_img0 = d0
            else:
                rows = (T - 1) * B
                db1 = colsum(dgx1, T * B, H4, H4)
                dW_hh1, dW_ih1, dW_hh0 = torch.zeros_like(w_hh1), torch.zeros_like(w_ih1), torch.zeros_like(w_hh0)
                if T > 1:
                    gemm_raw(dgx1[1:], y1[:-1], dW_hh1, H4, H, rows, 1, H4, H, 1, H, mode=mode, splitk=True)
                    gemm_raw(dgx0[1:], y0[:-1], dW_hh0, H4, H, rows, 1, H4, H, 1, H, mode=mode, splitk=True)
                gemm_raw(dgx1, y0, dW_ih1, H4, H, T * B, 1, H4, H, 1, H, mode=mode, splitk=True)
        if img_only:
            dgx0 = image_only_gradient((T, B, H4), dev, ctx.gx_dtype)
            _handoff_put_image_only(dgx0, d_img0)
        elif d_img0 is not None:
            if ctx.gx_dtype != torch.float32:
                dgx0 = dgx0.to(ctx.gx_dtype)
            _handoff_put(dgx0, d_img0)
        return dgx0, dW_hh0, dW_ih1, db1, db1, dW_hh1, None, None, None, None, None


def _pair_backward_sequential(ctx, dy1, w_img, img_only):
    """DecoderPairFn.backward without the pipeline: layer 1's recurrence, ONE dX GEMM over all valid rows, layer 0's recurrence -- the
    round-5 sequence (4-row kernels, dgates of both layers as compact images only).  The backward kernel at 8 rows per group and two
    roles runs 2.55 us per step against 2 x 1.62, and the chunked dX GEMMs (N = 1024, K = 4096: 224 tiles of a chunk) cost 3 x the one
    GEMM -- measured a loss of 0.25 ms per flow (profiles/r06_pair_pipeline.log); the forward pipeline stands on its own."""
    w_hh0, w_ih1, w_hh1, lens, y0, g0, c0, y1, g1, c1 = ctx.saved_tensors
    T, B, H = y1.shape
    H4, mode, rm = 4 * H, ctx.mode, ctx.rowmap
    dev = dy1.device
    st = _persist_watch(dev)
    code = PERSIST_BWD_CODE
    work = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=dev, dtype=torch.uint8)

    def recurrence(dy, g, c, w_hh):
        img = Bf16Image.empty_rows(H4, rm, mode, dev)
        L.check(L.op16("ft_lstm_persist_bwd_img", mode)(L.ptr(dy), H, L.ptr(w_hh), L.ptr(lens), L.ptr(g), L.ptr(c), None, L.ptr(work), L.ptr(st.status),
                                                        T, B, H, code, L.ptr(img.buf), img.ld, img.buf.numel() // (2 * img.ld), L.ptr(img.colsum), L.stream()),
                "ft_lstm_persist_bwd_img")
        _persist_arm(st)
        return img

    d_img1 = recurrence(dy1, g1, c1, w_hh1)
    y0_img, y1_img = y_image(y0, T * B, H, mode, rm), y_image(y1, T * B, H, mode, rm)
    dW_hh1, dW_ih1, dW_hh0 = weight_grad_out(w_hh1), weight_grad_out(w_ih1), weight_grad_out(w_hh0)
    gemm_img(d_img1, 1, d_img1.ptr(1), y1_img, 1, y1_img.ptr(0), dW_hh1, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2, k_shift=1)
    dy0 = torch.empty(T, B, H, device=dev, dtype=torch.float32)
    gemm_img(d_img1, 0, d_img1.ptr(), w_img, 1, w_img.ptr(), dy0, rm.cap, H, H4, H, rowmap=rm, compact=1)
    gemm_img(d_img1, 1, d_img1.ptr(), y0_img, 1, y0_img.ptr(), dW_ih1, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2)
    d_img0 = recurrence(dy0, g0, c0, w_hh0)
    gemm_img(d_img0, 1, d_img0.ptr(1), y0_img, 1, y0_img.ptr(0), dW_hh0, H4, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2, k_shift=1)
    dgx0 = image_only_gradient((T, B, H4), dev, ctx.gx_dtype)
    _handoff_put_image_only(dgx0, d_img0)
    db1 = d_img1.colsum
    return dgx0, dW_hh0, dW_ih1, db1, db1, dW_hh1, None, None, None, None, None


DecoderPairFn._backward_sequential = staticmethod(_pair_backward_sequential)


def decoder_pair(x, lens, p, mode, xs_extra, rowmap, fill, gate, nchunks):
    """the decoder nn.LSTM `p` (two layers) over [x ; xs_extra]: layer 0's input projection (with the gate layer riding on its image,
    lstm_layer) + DecoderPairFn; returns (h, gates)"""
    xs = [x] + list(xs_extra)
    T, B = x.shape[0], x.shape[1]
    gates = None
    if gx16_ok(mode, rowmap, T, B, p.weight_hh_l0.shape[1], False, xs, p.weight_ih_l0.shape[0], x.device):
        fill = fill + "|c16"
    if gate is not None and linear_gate_fusable(mode, rowmap, xs, p.weight_ih_l0.shape[0]):
        gx, gates = LinearGateFn.apply(p.weight_ih_l0, bias_sum(p.bias_ih_l0, p.bias_hh_l0), gate[0], gate[1], mode, rowmap, fill, *xs)
    else:
        gx = LinearFn.apply(p.weight_ih_l0, bias_sum(p.bias_ih_l0, p.bias_hh_l0), L.ACT_NONE, mode, rowmap, fill, *xs)
        if gate is not None:
            gates = linear(xs, gate[0], gate[1], mode=mode)
    private = rowmap is not None and rowmap.T == T and rowmap.B == B and linear_uses_images(mode, T * B, p.weight_ih_l0.shape[0], xs)
    h = DecoderPairFn.apply(gx, p.weight_hh_l0, p.weight_ih_l1, p.bias_ih_l1, p.bias_hh_l1, p.weight_hh_l1, lens, mode, rowmap, private, nchunks)
    return h, gates


class LSTMSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, w_hh, lens, reverse, mode, rowmap=None, gx_private=False):
        """gx_private: gx comes straight from a LinearFn over the same rowmap that differentiates through images (lstm_layer), so
        its gradient may be handed over as the 16-bit image alone"""
        ctx.gx_private = bool(gx_private)
        gx, w_hh = _c(gx), _c(w_hh)
        L.require_cuda(gx, w_hh, lens)
        T, B, H4 = gx.shape
        ctx.gx_dtype = gx.dtype                       # fp32, or 16-bit rows (lstm_layer: gx16_ok) for the persistent forward kernel
        if gx.dtype != torch.float32 and not (lstm_persist_groups(B, H4 // 4, reverse, mode, gx.device) or lstm_persist_slices(B, H4 // 4, reverse, mode, gx.device)):
            gx = gx.float()                           # (the device left the persistent kernels since the projection ran)
        ctx.rowmap = rowmap if (rowmap is not None and not reverse and rowmap.T == T and rowmap.B == B) else None
        H = H4 // 4
        y = torch.empty(T, B, H, device=gx.device, dtype=torch.float32)
        gates = torch.empty(T, B, H4, device=gx.device, dtype=torch.float32)
        cell = torch.empty(T, B, H, device=gx.device, dtype=torch.float32)
        if lstm_persist_groups(B, H, reverse, mode, gx.device) or lstm_persist_slices(B, H, reverse, mode, gx.device):
            # persistent forward recurrence (csrc/lstm_roles.hip, round 6): rows per XCD group follow the batch -- B <= 32: 4 rows, ONE
            # launch per sequence (1.62 us per step; the round-5 kernel it replaced: 1.76), B <= 64: 8 rows in one launch (1.93 against
            # 2 x 1.76 for two slices), wider: slices of 128 rows at 16 per group (2.77 against 4 x 1.76).  Bit-identical to the
            # launch-per-step kernel for every geometry.
            wimg = roles_wimg(w_hh, mode, False)
            for b0, nb, R in roles_plan(B, False):
                roles_launch([fwd_role(gx, lens, y, gates, cell, wimg, b0=b0, nb=nb)], R, mode, gx.device)
        else:
            work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=gx.device, dtype=torch.uint8)
            L.check(L.lib().ft_lstm_seq_fwd(L.ptr(gx), L.ptr(w_hh), L.ptr(lens), L.ptr(y), H, L.ptr(gates), L.ptr(cell),
                                            L.ptr(work), T, B, H, int(reverse), mode, L.stream()), "ft_lstm_seq_fwd")
        ctx.save_for_backward(w_hh, lens, y, gates, cell)
        ctx.reverse, ctx.mode = bool(reverse), mode
        return y

    @staticmethod
    def backward(ctx, dy):
        w_hh, lens, y, gates, cell = ctx.saved_tensors
        dy = _c(dy)
        T, B, H = y.shape
        dgx = None                                   # allocated below unless the gradient leaves as an image only
        ng = lstm_persist_groups(B, H, ctx.reverse, ctx.mode, dy.device)
        wide = (not ng) and lstm_persist_slices(B, H, ctx.reverse, ctx.mode, dy.device)
        d_img_k, img_only = None, False
        if wide:
            # slices of 64 rows at 8 per XCD group (2.5 us per step against 2 x 1.65 for two 32-row launches; 16 rows per group lose in the
            # backward kernel: 7.2 us per step for 128 rows, profiles/r06_persist_rows_per_group.log)
            dgx = torch.empty(T, B, 4 * H, device=dy.device, dtype=torch.float32)
            wimg = roles_wimg(w_hh, ctx.mode, True)
            for b0, nb, R in roles_plan(B, True):
                roles_launch([bwd_role(dy, lens, gates, cell, dgx, wimg, b0=b0, nb=nb)], R, ctx.mode, dy.device, backward=True)
        elif ng:
            # B <= 32: the REDUCE-SCATTER kernel (csrc/lstm_persist.hip, round 4: every CU multiplies its own dgates, fp32 partials cross
            # the XCD's L2; 1.62 us per step -- the roles backward kernel at 4 rows measures 1.71, so this one stays)
            ng = PERSIST_BWD_CODE
            st = _persist_watch(dy.device)
            work = torch.empty(L.lib().ft_lstm_persist_workspace_bytes(B, H), device=dy.device, dtype=torch.uint8)
            rm = ctx.rowmap
            if (_PERSIST_IMG != "0" and rm is not None and ctx.needs_input_grad[1] and T > 1 and images_apply(ctx.mode, 4 * H, H, (T - 1) * B)
                    and (ctx.gx_private or _PERSIST_IMG == "both")):
                # the output waves leave the compact 16-bit image of dgates and its column sums INSTEAD of the fp32 rows (beside
                # them with FLOWTRON_LSTM_PERSIST_IMG=both: 3.07 vs 2.91 us per step): no 450 MB dgx, no conversion pass over it.
                # Anomaly mode inspects every gradient a node returns: it gets real values (both), not the NaN face of an image.
                d_img_k = Bf16Image.empty_rows(4 * H, rm, ctx.mode, dy.device)
                img_only = _PERSIST_IMG != "both" and not torch.is_anomaly_enabled()
                if not img_only:
                    dgx = torch.empty(T, B, 4 * H, device=dy.device, dtype=torch.float32)
                L.check(L.op16("ft_lstm_persist_bwd_img", ctx.mode)(L.ptr(dy), H, L.ptr(w_hh), L.ptr(lens), L.ptr(gates), L.ptr(cell),
                                                    None if img_only else L.ptr(dgx),
                                                    L.ptr(work), L.ptr(st.status), T, B, H, ng, L.ptr(d_img_k.buf), d_img_k.ld,
                                                    d_img_k.buf.numel() // (2 * d_img_k.ld), L.ptr(d_img_k.colsum), L.stream()),
                        "ft_lstm_persist_bwd_img")
            else:
                dgx = torch.empty(T, B, 4 * H, device=dy.device, dtype=torch.float32)
                L.check(L.op16("ft_lstm_persist_bwd", ctx.mode)(L.ptr(dy), H, L.ptr(w_hh), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx),
                                                    L.ptr(work), L.ptr(st.status), T, B, H, ng, L.stream()), "ft_lstm_persist_bwd")
            _persist_arm(st)
        else:
            dgx = torch.empty(T, B, 4 * H, device=dy.device, dtype=torch.float32)
            work = torch.empty(L.lib().ft_lstm_workspace_bytes(B, H), device=dy.device, dtype=torch.uint8)
            L.check(L.lib().ft_lstm_seq_bwd(L.ptr(dy), H, L.ptr(w_hh), L.ptr(lens), L.ptr(gates), L.ptr(cell), L.ptr(dgx),
                                            L.ptr(work), T, B, H, int(ctx.reverse), ctx.mode, L.stream()), "ft_lstm_seq_bwd")
        if img_only:
            # the gradient autograd carries is the NaN face of the image: a foreign reader sees NaN, never unwritten memory
            dgx = image_only_gradient((T, B, 4 * H), dy.device, ctx.gx_dtype)
        dgx_f32 = dgx
        if not img_only and ctx.gx_dtype != torch.float32:
            # 16-bit gx rows (gx16_ok) in a pass that keeps fp32 dgates (anomaly mode, FLOWTRON_LSTM_PERSIST_IMG=both): autograd wants
            # the gradient in gx's dtype -- cast here, so that the image hand-off below is keyed on the tensor it will carry
            dgx = dgx.to(ctx.gx_dtype)
        dW = None
        if ctx.needs_input_grad[1]:
            # dW_hh[r,j] = sum_{t,b} da_t[b,r] * h_prev(t)[b,j];  h_prev = y[t-1] (fwd) / y[t+1] (reverse)
            rows = (T - 1) * B
            dW = weight_grad_out(w_hh)
            rm = ctx.rowmap
            if T > 1 and images_apply(ctx.mode, 4 * H, H, rows) and rm is not None:
                # compact images (valid frames only, batch-major with one zero separator row per utterance): the one-step shift
                # dgates_t <-> h_{t-1} is a shift by ONE compact row, and the utterance boundaries multiply with a separator
                d_img = d_img_k if d_img_k is not None else Bf16Image(dgx_f32.reshape(T * B, 4 * H), colsum=True, mode=ctx.mode, rowmap=rm)
                y_img = y_image(y, T * B, H, ctx.mode, rm)
                gemm_img(d_img, 1, d_img.ptr(1), y_img, 1, y_img.ptr(0), dW, 4 * H, H, rm.cap, H, beta=1.0, splitk=True, rowmap=rm, compact=2, k_shift=1)
                if img_only:
                    _handoff_put_image_only(dgx, d_img)     # ... and ONLY the image exists
                else:
                    _handoff_put(dgx, d_img)    # the input projection's backward reads the same dgates
            elif T > 1 and images_apply(ctx.mode, 4 * H, H, rows):
                # images of dgates / outputs over all T*B rows; the one-step shift is a row offset into them
                d_img, y_img = Bf16Image(dgx_f32.reshape(T * B, 4 * H), colsum=True, mode=ctx.mode), shared_image(y, T * B, H, ctx.mode)
                fwd = not ctx.reverse
                gemm_img(d_img, 1, d_img.ptr(B if fwd else 0), y_img, 1, y_img.ptr(0 if fwd else B), dW, 4 * H, H, rows, H, beta=1.0, splitk=True)
                _handoff_put(dgx, d_img)        # the input projection's backward reads the same dgates
            elif T > 1:
                da = dgx_f32[1:] if not ctx.reverse else dgx_f32[:-1]
                hp = y[:-1] if not ctx.reverse else y[1:]
                gemm_raw(da, hp, dW, 4 * H, H, rows, 1, 4 * H, H, 1, H, mode=ctx.mode, splitk=True)
        return dgx, dW, None, None, None, None, None


MAX_STEP_BATCH = 64          # batch rows the launch-per-step recurrences and the bidirectional pair chain take (csrc/lstm.hip)


def batch_chunks(B, cap):
    return [(b0, min(cap, B - b0)) for b0 in range(0, B, cap)]


def lstm_layer(x, lens, w_ih, w_hh, b_ih, b_hh, reverse=False, mode=None, xs_extra=None, rowmap=None, fill="", gate=None):
    """One LSTM layer over a padded sequence: input projection for all T*B rows (valid rows only with a RowMap) as one MFMA
    GEMM, then the sequential recurrence.  The recurrence kernels never use a padded frame's gx / dy row and write zeros to its
    y / dgx row themselves, so the projection leaves padded rows unwritten (fill = ""); pass fill = "dx" when something that
    reduces over all T frames reads the INPUT gradients (the attention context).
    gate = (weight [1, K], bias | None): also returns the N = 1 projection of the SAME inputs (the gate layer, flowtron.py:760-761)
    as (h, gates) -- computed from this projection's operand image where that exists."""
    mode = L.mfma_mode() if mode is None else mode
    xs = [x] if xs_extra is None else [x] + list(xs_extra)
    if reverse:
        rowmap = None
    T, B = x.shape[0], x.shape[1]
    gates = None
    H = w_hh.shape[1]
    if lstm_pad_width(B, H, reverse, mode, x.device, T):
        # a hidden size below the persistent kernels' 1024 (the reference's nn.LSTM takes any, flowtron.py:654-655): the recurrence of a
        # layer whose gate blocks are zero-padded to 1024 units IS this layer's recurrence -- a padded unit has gx = 0 and zero weights,
        # so its cell stays 0 (c = 0.5 c + 0.5 * 0) and its output 0 (0.5 tanh(0)) in every step, and the real units see zeros through
        # zero weight columns.  The padding is torch ops on the parameters: autograd slices the gradients back.
        w_ih_p, w_hh_p, b_p = pad_gate_blocks(w_ih, 0), pad_gate_blocks(w_hh, PERSIST_H - H), pad_gate_blocks(bias_sum(b_ih, b_hh), 0)
        out = lstm_layer(x, lens, w_ih_p, w_hh_p, b_p, torch.zeros_like(b_p), reverse, mode, xs_extra, rowmap, fill, gate)
        return out[..., :H].contiguous() if gate is None else (out[0][..., :H].contiguous(), out[1])
    if gx16_ok(mode, rowmap, T, B, H, reverse, xs, w_ih.shape[0], x.device):
        fill = fill + "|c16"                   # gx as 16-bit rows: half the bytes the projection writes and the recurrence reads
    if gate is not None and linear_gate_fusable(mode, rowmap, xs, w_ih.shape[0]):
        # gate = (weight [1, K], bias): the N = 1 projection over the same inputs rides on this projection's image (LinearGateFn)
        gx, gates = LinearGateFn.apply(w_ih, bias_sum(b_ih, b_hh), gate[0], gate[1], mode, rowmap, fill, *xs)
    else:
        gx = LinearFn.apply(w_ih, bias_sum(b_ih, b_hh), L.ACT_NONE, mode, rowmap, fill, *xs)
        if gate is not None:
            gates = linear(xs, gate[0], gate[1], mode=mode)
    private = rowmap is not None and rowmap.T == T and rowmap.B == B and linear_uses_images(mode, T * B, w_ih.shape[0], xs)
    if B > MAX_STEP_BATCH and not lstm_persist_slices(B, H, reverse, mode, gx.device):
        # the launch-per-step kernels take at most 64 batch rows (the reference's nn.LSTM takes any, flowtron.py:654-655): the rows
        # are independent, so the recurrence runs per batch chunk on contiguous copies and autograd splits the gradients again
        h = torch.cat([LSTMSeqFn.apply(gx[:, b0:b0 + nb].contiguous(), w_hh, lens[b0:b0 + nb], reverse, mode, None, False)
                       for b0, nb in batch_chunks(B, MAX_STEP_BATCH)], 1)
    else:
        h = LSTMSeqFn.apply(gx, w_hh, lens, reverse, mode, rowmap, private)
    return h if gate is None else (h, gates)


class BiLSTMSeqFn(torch.autograd.Function):
    """[h_fwd | h_rev] of a bidirectional layer, both recurrences in one launch chain (csrc/lstm.hip lstm_*_pair)."""

    @staticmethod
    def forward(ctx, gx_f, gx_r, w_f, w_r, lens, mode=L.FT_BF16):
        ctx.mode = mode
        gx_f, gx_r, w_f, w_r = _c(gx_f), _c(gx_r), _c(w_f), _c(w_r)
        L.require_cuda(gx_f, gx_r, w_f, w_r, lens)
        T, B, H4 = gx_f.shape
        H = H4 // 4
        f = dict(device=gx_f.device, dtype=torch.float32)
        y = torch.empty(T, B, 2 * H, **f)
        gates = [torch.empty(T, B, H4, **f) for _ in range(2)]
        cell = [torch.empty(T, B, H, **f) for _ in range(2)]
        if bilstm_persist_ok(B, H, mode, gx_f.device):
            st = _persist_watch(gx_f.device)
            work = torch.empty(L.lib().ft_bilstm_persist_workspace_bytes(B, H), device=gx_f.device, dtype=torch.uint8)
            L.check(L.op16("ft_bilstm_persist_fwd", mode)(L.ptr(gx_f), L.ptr(gx_r), L.ptr(w_f), L.ptr(w_r), L.ptr(lens), L.ptr(y), 2 * H,
                                                  L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]),
                                                  L.ptr(work), L.ptr(st.status), T, B, H, L.stream()), "ft_bilstm_persist_fwd")
            _persist_arm(st, bilstm=True)
        else:
            nb = L.lib().ft_lstm_workspace_bytes(B, H)
            work = [torch.empty(nb, device=gx_f.device, dtype=torch.uint8) for _ in range(2)]
            L.check(L.op16("ft_lstm_bidir_seq_fwd", mode)(L.ptr(gx_f), L.ptr(gx_r), L.ptr(w_f), L.ptr(w_r), L.ptr(lens), L.ptr(y), 2 * H,
                                                  L.ptr(gates[0]), L.ptr(gates[1]), L.ptr(cell[0]), L.ptr(cell[1]),
                                                  L.ptr(work[0]), L.ptr(work[1]), T, B, H, L.stream()), "ft_lstm_bidir_seq_fwd")
        ctx.save_for_backward(w_f, w_r, lens, y, gates[0], gates[1], cell[0], cell[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        w_f, w_r, lens, y, g0, g1, c0, c1 = ctx.saved_tensors
        dy = _c(dy)
        T, B, H2 = y.shape
        H = H2 // 2
        f = dict(device=dy.device, dtype=torch.float32)
        dgx = [torch.empty(T, B, 4 * H, **f) for _ in range(2)]
        if bilstm_persist_ok(B, H, ctx.mode, dy.device):
            st = _persist_watch(dy.device)
            work = torch.empty(L.lib().ft_bilstm_persist_workspace_bytes(B, H), device=dy.device, dtype=torch.uint8)
            L.check(L.op16("ft_bilstm_persist_bwd", ctx.mode)(L.ptr(dy), 2 * H, L.ptr(w_f), L.ptr(w_r), L.ptr(lens), L.ptr(g0), L.ptr(g1),
                                                  L.ptr(c0), L.ptr(c1), L.ptr(dgx[0]), L.ptr(dgx[1]), L.ptr(work), L.ptr(st.status),
                                                  T, B, H, L.stream()), "ft_bilstm_persist_bwd")
            _persist_arm(st, bilstm=True)
        else:
            nb = L.lib().ft_lstm_workspace_bytes(B, H)
            work = [torch.empty(nb, device=dy.device, dtype=torch.uint8) for _ in range(2)]
            L.check(L.op16("ft_lstm_bidir_seq_bwd", ctx.mode)(L.ptr(dy), 2 * H, L.ptr(w_f), L.ptr(w_r), L.ptr(lens), L.ptr(g0), L.ptr(g1),
                                                  L.ptr(c0), L.ptr(c1), L.ptr(dgx[0]), L.ptr(dgx[1]), L.ptr(work[0]), L.ptr(work[1]),
                                                  T, B, H, L.stream()), "ft_lstm_bidir_seq_bwd")
        dWs = [None, None]
        rows = (T - 1) * B
        for d in range(2):
            if ctx.needs_input_grad[2 + d]:
                dWs[d] = weight_grad_out(w_f if d == 0 else w_r)         # zero-filled: the GEMM accumulates (beta = 1)
                if T > 1:
                    # dW_hh[r,j] = sum da_t[b,r] h_prev(t)[b,j];  h_prev = y[t-1] (forward) / y[t+1] (reverse), y row stride 2H
                    da = dgx[d][1:] if d == 0 else dgx[d][:-1]
                    hp = y[:-1, :, :H] if d == 0 else y[1:, :, H:]
                    gemm_raw(da, hp, dWs[d], 4 * H, H, rows, 1, 4 * H, 2 * H, 1, H, beta=1.0, mode=ctx.mode, splitk=True)
        return dgx[0], dgx[1], dWs[0], dWs[1], None, None


def bilstm_layer(x, lens, wf, wr, mode=None):
    """Bidirectional LSTM layer.  wf / wr = (w_ih, w_hh, b_ih, b_hh) of the two directions.  Returns [T,B,2H]."""
    mode = L.mfma_mode() if mode is None else mode
    T, B, _ = x.shape
    H = wf[1].shape[1]
    import os
    if B > MAX_STEP_BATCH:
        # wider than the recurrence kernels take: batch chunks (independent rows; 32 rows each where the persistent bidirectional
        # kernels apply, else 64), concatenated again -- autograd splits the gradient
        cap = 32 if bilstm_persist_ok(32, H, mode, x.device) else MAX_STEP_BATCH
        return torch.cat([bilstm_layer(x[:, b0:b0 + nb], lens[b0:b0 + nb], wf, wr, mode) for b0, nb in batch_chunks(B, cap)], 1)
    if L.is16(mode) and os.environ.get("FLOWTRON_BILSTM", "1") != "0" and L.lib().ft_lstm_bidir_supported(B, H):
        gx_f = LinearFn.apply(wf[0], bias_sum(wf[2], wf[3]), L.ACT_NONE, mode, None, "", x)
        gx_r = LinearFn.apply(wr[0], bias_sum(wr[2], wr[3]), L.ACT_NONE, mode, None, "", x)
        return BiLSTMSeqFn.apply(gx_f, gx_r, wf[1], wr[1], lens, mode)
    yf = lstm_layer(x, lens, *wf, reverse=False, mode=mode)
    yb = lstm_layer(x, lens, *wr, reverse=True, mode=mode)
    return torch.cat([yf, yb], 2)


# --------------------------------------------------------------------------
# attention (flowtron.py:544-592)
# --------------------------------------------------------------------------
class AttentionScoresFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, v, in_lens, prior, temperature):
        Q, K, v = _c(Q), _c(K), _c(v.reshape(-1))
        L.require_cuda(Q, K, v, in_lens, prior)
        T, B, A = Q.shape
        Lk = K.shape[0]
        attn = torch.empty(B, T, Lk, device=Q.device, dtype=torch.float32)
        logprob = torch.empty_like(attn)
        p_save = None
        if prior is not None:
            prior = _c(prior.float())
            p_save = torch.empty_like(attn)
        L.check(L.lib().ft_attention_fwd(L.ptr(Q), L.ptr(K), L.ptr(v), L.ptr(in_lens), L.ptr(prior), L.ptr(attn), L.ptr(logprob),
                                         L.ptr(p_save), T, B, Lk, A, float(temperature), L.stream()), "ft_attention_fwd")
        ctx.save_for_backward(Q, K, v, in_lens, prior, attn, p_save)
        ctx.temperature = float(temperature)
        return attn, logprob

    @staticmethod
    def backward(ctx, dattn, dlogprob):
        Q, K, v, in_lens, prior, attn, p_save = ctx.saved_tensors
        T, B, A = Q.shape
        Lk = K.shape[0]
        if dattn is None:
            dattn = torch.zeros_like(attn)
        dattn = _c(dattn)
        if dlogprob is not None:
            dlogprob = _c(dlogprob)
        de = torch.empty_like(attn)
        dQ = torch.empty_like(Q)
        dK = torch.empty_like(K)
        dv = torch.zeros_like(v)
        L.check(L.lib().ft_attention_bwd(L.ptr(Q), L.ptr(K), L.ptr(v), L.ptr(in_lens), L.ptr(prior), L.ptr(attn), L.ptr(p_save),
                                         L.ptr(dattn), L.ptr(dlogprob), L.ptr(de), L.ptr(dQ), L.ptr(dK), L.ptr(dv),
                                         T, B, Lk, A, ctx.temperature, L.stream()), "ft_attention_bwd")
        return dQ, dK, dv.reshape(1, -1), None, None, None


class ContextFn(torch.autograd.Function):
    """ctx[t,b,:] = sum_l attn[b,t,l] V[l,b,:]  (torch.bmm at flowtron.py:590-591) as one batched MFMA GEMM."""

    @staticmethod
    def forward(ctx, attn, V, mode):
        attn, V = _c(attn), _c(V)
        L.require_cuda(attn, V)
        B, T, Lk = attn.shape
        A = V.shape[2]
        out = torch.empty(T, B, A, device=V.device, dtype=torch.float32)
        gemm_raw(attn, V, out, T, A, Lk, Lk, 1, B * A, 1, B * A, batch=B, bsA=T * Lk, bsB=A, bsC=A, mode=mode)
        ctx.save_for_backward(attn, V)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dctx):
        attn, V = ctx.saved_tensors
        dctx = _c(dctx)
        B, T, Lk = attn.shape
        A = V.shape[2]
        dattn = torch.empty_like(attn)
        # dattn[b,t,l] = sum_a dctx[t,b,a] V[l,b,a]
        gemm_raw(dctx, V, dattn, T, Lk, A, B * A, 1, 1, B * A, Lk, batch=B, bsA=A, bsB=A, bsC=T * Lk, mode=ctx.mode)
        dV = torch.empty_like(V)
        # dV[l,b,a] = sum_t attn[b,t,l] dctx[t,b,a]
        gemm_raw(attn, dctx, dV, Lk, A, T, 1, Lk, B * A, 1, B * A, batch=B, bsA=T * Lk, bsB=A, bsC=A, mode=ctx.mode)
        return dattn, dV, None


class CummAttnSeqFn(torch.autograd.Function):
    """run_cumm_attn_sequence (flowtron.py:697-723) as ONE pair of C-ABI calls per flow: the library walks the T dependent frames
    (csrc/cumm_attn.hip: location convolutions, key modulation, per-frame key projection, scores + softmax + context + running
    sum; 8 launches per frame forward, 22 backward, no Python / allocator / autograd work per frame).
    Q [T,B,A] and V [L,B,A] are the projected queries / values, text [L,B,E] the encoder outputs the keys are made from."""

    @staticmethod
    def forward(ctx, Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, temperature, mode):
        Q, V, text, w_key, v, w1, b1, w2, b2 = (_c(t) for t in (Q, V, text, w_key, v.reshape(-1), w1, b1, w2, b2))
        L.require_cuda(Q, V, text, w_key, v, w1, b1, w2, b2, in_lens)
        T, B, A = Q.shape
        Lk, _, E = text.shape
        NF, _, K1 = w1.shape
        K2 = w2.shape[2]
        f = dict(device=Q.device, dtype=torch.float32)
        out_ctx, attn, logprob = torch.empty(T, B, A, **f), torch.empty(B, T, Lk, **f), torch.empty(B, T, Lk, **f)
        cumm_all, kproj_all = torch.empty(T, B, Lk, **f), torch.empty(T, Lk * B, A, **f)
        work = torch.empty(L.lib().ft_cumm_attn_workspace_bytes(T, Lk, B, E, A, NF, K1, K2, int(mode), 0) + 256, device=Q.device, dtype=torch.uint8)
        args = CummAttnSeqFn._args(Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, out_ctx, attn, logprob, cumm_all, kproj_all, work,
                                   temperature, mode)
        # the fused frames as persistent launches (one for all forward frames, one per chunk of backward frames) on a device whose
        # whole-chip grids passed the self-test: same status word, same watch / poison / fallback chain as the persistent recurrences
        st = _persist_watch(Q.device) if (L.is16(mode) and _os.environ.get("FLOWTRON_CUMM_PERSIST", "1") != "0" and persist_usable(Q.device)) else None
        if st is not None:
            args.persist_status = L.ptr(st.status)
        L.check(L.lib().ft_cumm_attn_fwd(C.byref(args), L.stream()), "ft_cumm_attn_fwd")
        if st is not None and L.lib().ft_cumm_attn_fused(C.byref(args)):
            _persist_arm(st)
        ctx.save_for_backward(Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, attn, cumm_all, kproj_all)
        ctx.temperature, ctx.mode = float(temperature), mode
        return out_ctx, attn, logprob

    @staticmethod
    def _args(Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, out_ctx, attn, logprob, cumm_all, kproj_all, work, temperature, mode):
        T, B, A = Q.shape
        Lk, _, E = text.shape
        base = (work.data_ptr() + 255) // 256 * 256
        return L.CummAttnArgs(L.ptr(text), L.ptr(Q), L.ptr(V), L.ptr(w_key), L.ptr(v), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2),
                              L.ptr(in_lens), L.ptr(out_ctx), L.ptr(attn), L.ptr(logprob), L.ptr(cumm_all), L.ptr(kproj_all),
                              base, work.numel() - (base - work.data_ptr()), T, B, Lk, E, A, w1.shape[0], w1.shape[2], w2.shape[2],
                              float(temperature), int(mode), None)

    @staticmethod
    def backward(ctx, dctx, dattn, dlogprob):
        Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, attn, cumm_all, kproj_all = ctx.saved_tensors
        T, B, A = Q.shape
        Lk, _, E = text.shape
        NF, _, K1 = w1.shape
        K2 = w2.shape[2]
        f = dict(device=Q.device, dtype=torch.float32)
        dctx = _c(dctx) if dctx is not None else torch.zeros(T, B, A, **f)
        dattn = _c(dattn) if dattn is not None else None
        dlogprob = _c(dlogprob) if dlogprob is not None else None
        dQ, dV, dtext = torch.empty_like(Q), torch.empty_like(V), torch.empty_like(text)
        dwk, dv, dw1, db1, dw2, db2 = (torch.empty_like(t) for t in (w_key, v, w1, b1, w2, b2))
        work = torch.empty(L.lib().ft_cumm_attn_workspace_bytes(T, Lk, B, E, A, NF, K1, K2, int(ctx.mode), 1) + 256, device=Q.device, dtype=torch.uint8)
        scratch = torch.empty(1, **f)                 # forward-only outputs are not written by the backward call
        args = CummAttnSeqFn._args(Q, V, text, w_key, v, w1, b1, w2, b2, in_lens, scratch, attn, scratch, cumm_all, kproj_all, work,
                                   ctx.temperature, ctx.mode)
        L.check(L.lib().ft_cumm_attn_bwd(C.byref(args), L.ptr(dctx), L.ptr(dattn), L.ptr(dlogprob), L.ptr(dQ), L.ptr(dV), L.ptr(dtext),
                                         L.ptr(dwk), L.ptr(dv), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.stream()), "ft_cumm_attn_bwd")
        return dQ, dV, dtext, dwk, dv.reshape(1, -1), dw1, db1, dw2, db2, None, None, None


# --------------------------------------------------------------------------
# affine coupling, reverse-by-length
# --------------------------------------------------------------------------
class AffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, x):
        out, x = _c(out), _c(x)
        L.require_cuda(out, x)
        M = x.shape[-1]
        rows = x.numel() // M
        z = torch.empty_like(x)
        L.check(L.lib().ft_affine_fwd(L.ptr(out), L.ptr(x), L.ptr(z), rows, M, L.stream()), "ft_affine_fwd")
        ctx.save_for_backward(out, x)
        return z

    @staticmethod
    def backward(ctx, dz):
        out, x = ctx.saved_tensors
        dz = _c(dz)
        M = x.shape[-1]
        rows = x.numel() // M
        dout = torch.empty_like(out)
        dx = torch.empty_like(x)
        L.check(L.lib().ft_affine_bwd(L.ptr(out), L.ptr(x), L.ptr(dz), None, L.ptr(dout), L.ptr(dx), rows, M, L.stream()),
                "ft_affine_bwd")
        return dout, dx


class ReverseByLengthFn(torch.autograd.Function):
    """flip + per-sample roll (flowtron.py:606-613) in closed form; an involution, so backward is the same gather."""

    @staticmethod
    def forward(ctx, x, lens, time_major):
        x = _c(x)
        L.require_cuda(x, lens)
        if time_major:
            T, B, Cc = x.shape
        else:
            B, T, Cc = x.shape
        y = torch.empty_like(x)
        L.check(L.lib().ft_reverse_by_length(L.ptr(x), L.ptr(y), L.ptr(lens), T, B, Cc, int(time_major), L.stream()),
                "ft_reverse_by_length")
        ctx.save_for_backward(lens)
        ctx.time_major = time_major
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        return ReverseByLengthFn.apply(_c(dy), lens, ctx.time_major), None, None


def reverse_by_length(x, lens, time_major=True):
    return ReverseByLengthFn.apply(x, lens, time_major)


# --------------------------------------------------------------------------
# losses (flowtron.py:200-243)
# --------------------------------------------------------------------------
class NLLFn(torch.autograd.Function):
    """[sum m z^2/(2 sigma^2) - sum_f sum m log_s_f] / (n_valid_frames * n_mel).
    log_s tensors are strided views [T,B,M] of the coupling output [T,B,2M] (row stride 2M)."""

    @staticmethod
    def forward(ctx, z, lens, sigma, *log_s):
        z = _c(z)
        L.require_cuda(z, lens)
        T, B, M = z.shape
        acc = torch.zeros(2, device=z.device, dtype=torch.float32)
        st = L.stream()
        L.check(L.lib().ft_masked_sum(L.ptr(z), M, L.ptr(lens), L.ptr(acc), 1, T, B, M, st), "ft_masked_sum")
        lds = []
        for ls in log_s:
            assert ls.shape == z.shape and ls.stride(2) == 1 and ls.stride(1) * B == ls.stride(0)
            lds.append(ls.stride(1))
            L.check(L.lib().ft_masked_sum(L.ptr(ls), ls.stride(1), L.ptr(lens), L.ptr(acc[1:]), 0, T, B, M, st), "ft_masked_sum")
        n = lens_total(lens, float(M))
        nll = (acc[0] / (2.0 * sigma * sigma) - acc[1]) / n
        ctx.save_for_backward(z, lens, n)
        ctx.sigma, ctx.n_ls = sigma, len(log_s)
        return nll

    @staticmethod
    def backward(ctx, g):
        z, lens, n = ctx.saved_tensors
        T, B, M = z.shape
        scale = (g.reshape(1).to(torch.float32) / n).contiguous()
        st = L.stream()
        dz = torch.empty_like(z)
        L.check(L.lib().ft_masked_sum_bwd(L.ptr(z), M, L.ptr(lens), L.ptr(scale), 1.0 / (ctx.sigma * ctx.sigma), 1,
                                          L.ptr(dz), M, T, B, M, st), "ft_masked_sum_bwd")
        dls = None
        if ctx.n_ls:
            dls = torch.empty_like(z)
            L.check(L.lib().ft_masked_sum_bwd(None, M, L.ptr(lens), L.ptr(scale), -1.0, 0, L.ptr(dls), M, T, B, M, st),
                    "ft_masked_sum_bwd")
        return (dz, None, None) + (dls,) * ctx.n_ls


class GateBCEFn(torch.autograd.Function):
    """sum_valid BCEWithLogits(gate, target) / n_valid_frames (flowtron.py:237-243). gate [T,B,1], target [B,T]."""

    @staticmethod
    def forward(ctx, gate, target, lens):
        gate, target = _c(gate), _c(target.float())
        L.require_cuda(gate, target, lens)
        T, B = gate.shape[0], gate.shape[1]
        acc = torch.zeros(1, device=gate.device, dtype=torch.float32)
        L.check(L.lib().ft_gate_bce_fwd(L.ptr(gate), L.ptr(target), L.ptr(lens), L.ptr(acc), T, B, L.stream()), "ft_gate_bce_fwd")
        n = lens_total(lens)
        ctx.save_for_backward(gate, target, lens, n)
        return (acc / n).reshape(())

    @staticmethod
    def backward(ctx, g):
        gate, target, lens, n = ctx.saved_tensors
        T, B = gate.shape[0], gate.shape[1]
        scale = (g.reshape(1).to(torch.float32) / n).contiguous()
        dgate = torch.empty_like(gate)
        L.check(L.lib().ft_gate_bce_bwd(L.ptr(gate), L.ptr(target), L.ptr(lens), L.ptr(scale), 1.0, L.ptr(dgate), T, B, L.stream()),
                "ft_gate_bce_bwd")
        return dgate, None, None


# --------------------------------------------------------------------------
# beta-binomial attention prior on device (data.py:31-41; the reference spends ~0.45 s/utterance in scipy)
# --------------------------------------------------------------------------
def beta_binomial_prior(in_lens, out_lens, T=None, Lk=None, scaling=1.0):
    """-> [B,T,L] fp32 prior for a whole batch, zero padded like DataCollate (data.py:225-243)."""
    L.require_cuda(in_lens, out_lens)
    i32, o32 = lens32(in_lens), lens32(out_lens)
    B = i32.numel()
    if T is None:
        T = int(out_lens.max())
    if Lk is None:
        Lk = int(in_lens.max())
    prior = torch.empty(B, T, Lk, device=i32.device, dtype=torch.float32)
    L.check(L.lib().ft_beta_binomial_prior(L.ptr(i32), L.ptr(o32), L.ptr(prior), B, T, Lk, float(scaling), L.stream()),
            "ft_beta_binomial_prior")
    return prior


# --------------------------------------------------------------------------
# elementwise add / mul (cumulative-attention branch, flowtron.py:712, :719)
# --------------------------------------------------------------------------
def _elt(a, b, op):
    out = torch.empty_like(a)
    L.check(L.lib().ft_eltwise(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), op, L.stream()), "ft_eltwise")
    return out


class MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        L.require_cuda(a, b)
        assert a.shape == b.shape
        ctx.save_for_backward(a, b)
        return _elt(a, b, 1)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        return (_elt(g, b, 1) if ctx.needs_input_grad[0] else None), (_elt(g, a, 1) if ctx.needs_input_grad[1] else None)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        L.require_cuda(a, b)
        assert a.shape == b.shape
        return _elt(a, b, 0)

    @staticmethod
    def backward(ctx, g):
        return g, g


# --------------------------------------------------------------------------
# attention-CTC loss (flowtron.py:155-182): banded DP kernel, one workgroup per sample
# --------------------------------------------------------------------------
class AttnCTCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, in_lens32, out_lens32, blank_logprob):
        lp = _c(lp.float())
        L.require_cuda(lp, in_lens32, out_lens32)
        B, T, Lk = lp.shape
        work = torch.empty(L.lib().ft_attn_ctc_workspace_floats(B, T, Lk), device=lp.device, dtype=torch.float32)
        loss = torch.empty(1, device=lp.device, dtype=torch.float32)
        ctx.with_beta = int(bool(ctx.needs_input_grad[0]))       # beta recursion beside alpha, in the same launch
        L.check(L.lib().ft_attn_ctc_fwd(L.ptr(lp), L.ptr(in_lens32), L.ptr(out_lens32), float(blank_logprob), L.ptr(work),
                                        L.ptr(loss), B, T, Lk, ctx.with_beta, L.stream()), "ft_attn_ctc_fwd")
        ctx.save_for_backward(lp, in_lens32, out_lens32, work)
        ctx.blank = float(blank_logprob)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lp, in32, out32, work = ctx.saved_tensors
        B, T, Lk = lp.shape
        gd = g.reshape(1).to(torch.float32).contiguous()
        dlp = torch.empty_like(lp)
        L.check(L.lib().ft_attn_ctc_bwd(L.ptr(lp), L.ptr(in32), L.ptr(out32), ctx.blank, L.ptr(work), L.ptr(gd), L.ptr(dlp),
                                        B, T, Lk, ctx.with_beta, L.stream()), "ft_attn_ctc_bwd")
        return dlp, None, None, None


# --------------------------------------------------------------------------
# FlowtronLoss as ONE autograd node (flowtron.py:200-274): NLL + gate + attention-CTC of all flows
# --------------------------------------------------------------------------
FUSED_LOSS = True


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class FlowtronLossFn(torch.autograd.Function):
    """(nll, gate_loss, ctc) from seven launches, their gradients from three -- where NLLFn + GateBCEFn + AttnCTCFn and the torch
    arithmetic around them (frame counts, normalisers, `g / n`, the concatenation of the flows' log-probabilities, the flip of the
    back-step flows', `lens.repeat(F)`) took ~55 dispatches of 4-20 us per training step, most of them in front of the host's
    .item() reads (train.py:300-303) and at the head of backward, where the GPU waits for the host.
    tensors = n_ls log_s views [T,B,M] (strided views of the coupling outputs) followed by the flows' attention log-probabilities
    [B,T,L] in flows order, each in ITS flow's time order (odd flows reversed: ft_attn_ctc_fwd_multi mirrors the row index)."""

    @staticmethod
    def forward(ctx, z, gate_pred, gate_target, out32, in32, sigma, blank, n_ls, *tensors):
        z = _c(z)
        log_s, lps = list(tensors[:n_ls]), [_c(t) for t in tensors[n_ls:]]
        L.require_cuda(z, gate_pred, gate_target, out32, in32, *log_s, *lps)
        T, B, M = z.shape
        dev = z.device
        st = L.stream()
        ld = M
        for ls in log_s:
            assert ls.shape == z.shape and ls.dtype == torch.float32 and ls.stride(2) == 1 and ls.stride(1) * B == ls.stride(0)
        if log_s:
            ld = int(log_s[0].stride(1))
            assert all(int(ls.stride(1)) == ld for ls in log_s)
        acc = torch.empty(8, device=dev, dtype=torch.float32)
        nll = torch.empty((), device=dev, dtype=torch.float32)
        gate_loss = None
        if gate_pred is not None:
            gate_pred, gate_target = _c(gate_pred), _c(gate_target.float())
            assert gate_pred.numel() == T * B and gate_target.shape == (B, T)
            gate_loss = torch.empty((), device=dev, dtype=torch.float32)
        L.check(L.lib().ft_flowtron_loss_fwd(L.ptr(z), _ptr_array(log_s) if log_s else None, len(log_s), ld, L.ptr(gate_pred),
                                             L.ptr(gate_target), L.ptr(out32), float(sigma), L.ptr(acc), L.ptr(nll), L.ptr(gate_loss),
                                             T, B, M, st), "ft_flowtron_loss_fwd")
        ctc = work = None
        ctx.with_beta = 0
        if lps:
            F_, (Bl, Tl, Lk) = len(lps), lps[0].shape
            assert Bl == B and all(lp.shape == lps[0].shape and lp.dtype == torch.float32 for lp in lps)
            ctx.rev = (C.c_int32 * F_)(*[f % 2 for f in range(F_)])          # back-step flows: reversed time (flowtron.py:250-256)
            ctx.with_beta = int(any(ctx.needs_input_grad[8 + n_ls + f] for f in range(F_)))
            work = torch.empty(L.lib().ft_attn_ctc_workspace_floats(F_ * B, Tl, Lk), device=dev, dtype=torch.float32)
            ctc = torch.empty((), device=dev, dtype=torch.float32)
            L.check(L.lib().ft_attn_ctc_fwd_multi(_ptr_array(lps), ctx.rev, F_, L.ptr(in32), L.ptr(out32), float(blank), L.ptr(work),
                                                  L.ptr(ctc), B, Tl, Lk, ctx.with_beta, st), "ft_attn_ctc_fwd_multi")
        ctx.save_for_backward(z, gate_pred, gate_target, out32, in32, acc, work, *lps)
        ctx.sigma, ctx.blank, ctx.n_ls, ctx.n_lp = float(sigma), float(blank), n_ls, len(lps)
        outs = (nll, gate_loss if gate_loss is not None else torch.zeros(1, device=dev),
                ctc if ctc is not None else torch.zeros(1, device=dev))
        ctx.mark_non_differentiable(*[o for o, live in zip(outs[1:], (gate_loss is not None, ctc is not None)) if not live])
        return outs

    @staticmethod
    def backward(ctx, g_nll, g_gate, g_ctc):
        z, gate_pred, gate_target, out32, in32, acc, work, *lps = ctx.saved_tensors
        T, B, M = z.shape
        st = L.stream()

        def scalar(g):
            return None if g is None else _c(g.reshape(1).to(torch.float32))

        g_nll = scalar(g_nll) if (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[8:8 + ctx.n_ls])) else None
        g_gate = scalar(g_gate) if (gate_pred is not None and ctx.needs_input_grad[1]) else None
        dz = dls = dgate = None
        if g_nll is not None:
            dz = torch.empty_like(z)
            dls = torch.empty_like(z) if ctx.n_ls else None
        if g_gate is not None:
            dgate = torch.empty_like(gate_pred)
        if dz is not None or dgate is not None:
            L.check(L.lib().ft_flowtron_loss_bwd(L.ptr(z), L.ptr(gate_pred), L.ptr(gate_target), L.ptr(out32), ctx.sigma, L.ptr(acc),
                                                 L.ptr(g_nll), L.ptr(g_gate), L.ptr(dz), L.ptr(dls), L.ptr(dgate), T, B, M, st),
                    "ft_flowtron_loss_bwd")
        dlps = [None] * ctx.n_lp
        if lps and g_ctc is not None and any(ctx.needs_input_grad[8 + ctx.n_ls:]):
            g_ctc = scalar(g_ctc)
            dlps = [torch.empty_like(lp) for lp in lps]
            _, Tl, Lk = lps[0].shape
            L.check(L.lib().ft_attn_ctc_bwd_multi(_ptr_array(lps), ctx.rev, len(lps), L.ptr(in32), L.ptr(out32), ctx.blank, L.ptr(work),
                                                  L.ptr(g_ctc), _ptr_array(dlps), B, Tl, Lk, ctx.with_beta, st), "ft_attn_ctc_bwd_multi")
        return (dz, dgate, None, None, None, None, None, None) + (dls,) * ctx.n_ls + tuple(dlps)


# --------------------------------------------------------------------------
# two stacked LSTM layers as one launch chain (csrc/lstm2.hip)
# --------------------------------------------------------------------------
class LSTM2SeqFn(torch.autograd.Function):
    """y1 = LSTM_l1(LSTM_l0(gx0)); gx0 = x W_ih0^T + b0 comes from LinearFn.  bf16 MFMA operands, forward direction."""

    @staticmethod
    def forward(ctx, gx0, w_hh0, w_ih1, b_ih1, b_hh1, w_hh1, lens, mode=L.FT_BF16):
        ctx.mode = mode
        gx0, w_hh0, w_ih1, w_hh1 = _c(gx0), _c(w_hh0), _c(w_ih1), _c(w_hh1)
        L.require_cuda(gx0, w_hh0, w_ih1, w_hh1, lens)
        T, B, H4 = gx0.shape
        H = H4 // 4
        f = dict(device=gx0.device, dtype=torch.float32)
        y0, y1 = torch.empty(T, B, H, **f), torch.empty(T, B, H, **f)
        gates0, gates1 = torch.empty(T, B, H4, **f), torch.empty(T, B, H4, **f)
        cell0, cell1 = torch.empty(T, B, H, **f), torch.empty(T, B, H, **f)
        bias1 = (b_ih1 + b_hh1).contiguous()
        work = torch.empty(L.lib().ft_lstm2_workspace_bytes(B, H), device=gx0.device, dtype=torch.uint8)
        L.check(L.op16("ft_lstm2_seq_fwd", mode)(L.ptr(gx0), L.ptr(w_hh0), L.ptr(w_ih1), L.ptr(bias1), L.ptr(w_hh1), L.ptr(lens), L.ptr(y0),
                                         L.ptr(gates0), L.ptr(cell0), L.ptr(y1), L.ptr(gates1), L.ptr(cell1), L.ptr(work), T, B, H,
                                         L.stream()), "ft_lstm2_seq_fwd")
        ctx.save_for_backward(w_hh0, w_ih1, w_hh1, lens, y0, gates0, cell0, y1, gates1, cell1)
        return y1

    @staticmethod
    def backward(ctx, dy1):
        w_hh0, w_ih1, w_hh1, lens, y0, gates0, cell0, y1, gates1, cell1 = ctx.saved_tensors
        dy1 = _c(dy1)
        T, B, H = y1.shape
        dgx0 = torch.empty(T, B, 4 * H, device=dy1.device, dtype=torch.float32)
        dgx1 = torch.empty_like(dgx0)
        work = torch.empty(L.lib().ft_lstm2_workspace_bytes(B, H), device=dy1.device, dtype=torch.uint8)
        L.check(L.op16("ft_lstm2_seq_bwd", ctx.mode)(L.ptr(dy1), L.ptr(w_hh0), L.ptr(w_ih1), L.ptr(w_hh1), L.ptr(lens), L.ptr(gates0), L.ptr(cell0),
                                         L.ptr(gates1), L.ptr(cell1), L.ptr(dgx0), L.ptr(dgx1), L.ptr(work), T, B, H, L.stream()),
                "ft_lstm2_seq_bwd")
        mode = ctx.mode
        rows = T * B
        dW_hh0 = torch.zeros_like(w_hh0)
        dW_hh1 = torch.zeros_like(w_hh1)
        dW_ih1 = torch.empty_like(w_ih1)
        r1 = (T - 1) * B
        db1 = None
        if T > 1 and images_apply(mode, 4 * H, H, r1):
            # four images serve the three weight-gradient GEMMs (the one-step shift is a row offset) and, through the
            # hand-off, the dX / dW GEMMs of the layer-0 input projection
            d0, d1 = Bf16Image(dgx0.reshape(rows, 4 * H), colsum=True, mode=mode), Bf16Image(dgx1.reshape(rows, 4 * H), colsum=True, mode=mode)
            i0, i1 = Bf16Image(y0.reshape(rows, H), mode=mode), shared_image(y1, rows, H, mode)
            gemm_img(d0, 1, d0.ptr(B), i0, 1, i0.ptr(), dW_hh0, 4 * H, H, r1, H, splitk=True)
            gemm_img(d1, 1, d1.ptr(B), i1, 1, i1.ptr(), dW_hh1, 4 * H, H, r1, H, splitk=True)
            gemm_img(d1, 1, d1.ptr(), i0, 1, i0.ptr(), dW_ih1, 4 * H, H, rows, H, splitk=True)
            _handoff_put(dgx0, d0)
            db1 = d1.colsum
        else:
            if T > 1:
                gemm_raw(dgx0[1:], y0[:-1], dW_hh0, 4 * H, H, r1, 1, 4 * H, H, 1, H, mode=mode, splitk=True)
                gemm_raw(dgx1[1:], y1[:-1], dW_hh1, 4 * H, H, r1, 1, 4 * H, H, 1, H, mode=mode, splitk=True)
            gemm_raw(dgx1, y0, dW_ih1, 4 * H, H, rows, 1, 4 * H, H, 1, H, mode=mode, splitk=True)
        if db1 is None:
            db1 = colsum(dgx1, rows, 4 * H, 4 * H)
        return dgx0, dW_hh0, dW_ih1, db1, db1, dW_hh1, None, None


def lstm2_supported(B, H, mode):
    import os
    return (L.is16(mode) and os.environ.get("FLOWTRON_LSTM2", "1") != "0" and bool(L.lib().ft_lstm2_supported(B, H)))
