"""RAdam (reference radam.py:26-122) as TWO HIP kernels over the flat arenas -- a sum-of-squares
reduction for the global grad norm (train.py:328 clip_grad_norm_) and one fused clip + moment + update
pass -- instead of ~12 small kernels x 68 tensors.

Optimizer state keeps the reference's per-parameter keys (`step`, `exp_avg`, `exp_avg_sq`; radam.py:63-66)
as views into flat moment arenas, so `optimizer.state_dict()` stays checkpoint-compatible (train.py:123,138), and
`load_state_dict` copies a loaded checkpoint's moments back INTO the arenas (the kernels read only the arenas).

Parameters whose .grad is None at step() time are skipped like radam.py:57-58 (`if p.grad is None: continue`): no moment
update, no weight decay, their state['step'] does not advance; they do not enter the global norm either (torch
clip_grad_norm_ ignores them).  Deviation: bias correction and N_sma use ONE global step count for the whole arena (radam.py
keeps one per parameter; they only differ for a parameter that was without a gradient for some iterations).

Guard: the fused kernel drops the whole update when the global gradient norm is NaN / Inf (ft_radam_step): the device-side
equivalent of GradScaler's overflow skip (train.py:330), which also catches a step poisoned by a persistent recurrence that
reported a time-out (`poison_from_status`) -- on every rank, because the poison travels through the gradient all-reduce.
`skipped_steps` reads the device counter.  A dropped update leaves weights and moments untouched; the host-side step count (bias
correction, N_sma) is taken back by the number of dropped updates whenever the host next LOOKS at the counter -- `skipped_steps`,
`state_dict()` (every checkpoint) -- so a GradScaler-style skip does not advance the schedule for good, without a
host synchronisation per step; between two looks the step size of a run that dropped k updates is the one of step n + k.
"""
from __future__ import annotations

import math

import torch
from torch.optim.optimizer import Optimizer

import os

from . import _lib as L
from .dist import FlatArena

_DEV_STEP = True      # bias-correction step count formed on the device (calls - drops)


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, arena: FlatArena = None):
        params = list(params)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("the fused RAdam runs one flat arena = one param group")
        plist = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if arena is None:
            arena = FlatArena.for_params(plist, flatten_params=True)
        else:
            assert [id(p) for p in arena.params] == [id(p) for p in plist], "arena / optimizer parameter order differs"
        self.arena = arena
        self.flat_m = torch.zeros_like(arena.flat_grad)
        self.flat_v = torch.zeros_like(arena.flat_grad)
        self.gnorm_sq = torch.zeros(1, device=arena.flat_grad.device, dtype=torch.float32)
        self._partials = torch.empty(L.SUMSQ_PARTIALS, device=arena.flat_grad.device, dtype=torch.float32)
        self._skipped = torch.zeros(1, device=arena.flat_grad.device, dtype=torch.int32)
        self._have_norm = False
        self._step = 0
        self._skipped_applied = 0        # device-dropped updates already taken back from _step
        # torch.amp.GradScaler.step (train.py:330 `scaler.step(optimizer)`): an optimizer with this attribute is handed `found_inf` /
        # `grad_scale` as DEVICE tensors and called unconditionally, instead of the scaler reading found_inf on the host (one queue drain
        # per fp16 step) and skipping the call.  The fused kernel's guard drops the update exactly then (an Inf / NaN gradient makes
        # the global norm non-finite), and the step count the schedule uses is formed on the device from the drops
        # (ft_radam_step_dev), so the trajectory is the one of the skipped call.
        self._step_supports_amp_scaling = True     # (=1: torch's host-side skip, for A/B)
        self._bind_state()

    def _bind_state(self):
        """state[p] = views into the flat moment arenas (keys of radam.py:63-66)."""
        a = self.arena
        for p, off in zip(a.params, a.offsets):
            k = p.numel()
            self.state[p] = {"step": self._step, "exp_avg": self.flat_m[off:off + k].view_as(p.data),
                             "exp_avg_sq": self.flat_v[off:off + k].view_as(p.data)}

    def load_state_dict(self, state_dict):
        """train.py:123 resume: the base class swaps state[p] for fresh tensors; the fused kernel reads flat_m / flat_v /
        _step, so the loaded moments are copied into the arenas and state[p] is re-pointed at the views."""
        super().load_state_dict(state_dict)
        a = self.arena
        step = 0
        with torch.no_grad():
            for p, off in zip(a.params, a.offsets):
                st = self.state.get(p)
                k = p.numel()
                if st and "exp_avg" in st:
                    self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1).to(self.flat_m))
                    self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1).to(self.flat_v))
                    step = max(step, int(st.get("step", 0)))
                else:                                  # parameter never stepped in the saved run (radam.py:60-66 lazy init)
                    self.flat_m[off:off + k].zero_()
                    self.flat_v[off:off + k].zero_()
        self._step = step
        # the loaded count is of APPLIED updates: drops the device counter holds that the host has not accounted for yet belong to the
        # run before the load and must not be subtracted from it later
        self._skipped_applied = int(self._skipped.item())
        self._bind_state()

    @staticmethod
    def step_size_for(step, lr, beta1, beta2):
        """radam.py:82-106 (the 10-slot buffer there is only a cache of this closed form)."""
        beta2_t = beta2 ** step
        n_sma_max = 2 / (1 - beta2) - 1
        n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma >= 5:
            ss = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max /
                                (n_sma_max - 2)) / (1 - beta1 ** step)
            return ss, True
        return lr / (1 - beta1 ** step), False

    @property
    def skipped_steps(self) -> int:
        """updates the device-side guard dropped so far (non-finite global gradient norm); one host read.  The step count that
        drives bias correction / N_sma is taken back by the drops not yet accounted for (radam.py counts only updates it applied)."""
        n = int(self._skipped.item())
        if n > self._skipped_applied:
            self._step = max(0, self._step - (n - self._skipped_applied))
            self._skipped_applied = n
            for st in self.state.values():
                if isinstance(st, dict) and "step" in st:
                    st["step"] = min(int(st["step"]), self._step)
        return n

    def state_dict(self):
        _ = self.skipped_steps           # a checkpoint carries the step count of the updates that were APPLIED
        return super().state_dict()

    def _norm_sq(self):
        """||g||^2 of the arena on device; a persistent recurrence that reported a time-out poisons it first (NaN)."""
        a = self.arena
        a.adopt_stray_grads(copy=True)
        poison_from_status(a.flat_grad)
        self.gnorm_sq.zero_()
        L.check(L.lib().ft_sumsq(L.ptr(a.flat_grad), L.ptr(self.gnorm_sq), a.numel, L.ptr(self._partials), L.stream()), "ft_sumsq")
        self._have_norm = True
        self._norm_version = a.flat_grad._version      # autograd's in-place accumulation into the arena views bumps it

    def clip_grad_norm_(self, max_norm: float):
        """Enqueue ||g||^2 on device and remember the clip for the next step(); returns the device scalar ||g||^2
        (no host sync -- torch.nn.utils.clip_grad_norm_ at train.py:328 does 68 norms and a sync)."""
        self._norm_sq()
        self._clip = float(max_norm)
        return self.gnorm_sq

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        a = self.arena
        # gradients that changed since clip_grad_norm_ took the norm (another backward in between: in-place accumulation bumps the
        # arena's version, fresh p.grad tensors are copied in by the adoption below) need the norm again.  The version is
        # read BEFORE the adoption: zero-filling the slice of a parameter without a gradient bumps it too, and used to cost one extra
        # ft_sumsq over the 244 MB arena per step
        stale = not self._have_norm or a.flat_grad._version != getattr(self, "_norm_version", -1)
        skipped = a.adopt_stray_grads(copy=True)
        stale = stale or a.adopt_copied
        # radam.py:57-58: a parameter without a gradient is left untouched (rare: frozen / unused branches) -- the fused
        # kernel sweeps the whole arena, so its slices are restored afterwards
        keep = [(off, k, a.flat_param[off:off + k].clone(), self.flat_m[off:off + k].clone(), self.flat_v[off:off + k].clone())
                for off, k in skipped]
        self._step += 1
        beta1, beta2 = g["betas"]
        clip = getattr(self, "_clip", 0.0)
        # GradScaler.step on a `_step_supports_amp_scaling` optimizer: gradients still scaled (no scaler.unscale_ before) arrive with
        # their scale as a device tensor -- divided out here, on the device.  found_inf needs no look: a non-finite gradient makes the
        # norm below non-finite, and the kernel drops the update on that.
        grad_scale = getattr(self, "grad_scale", None)
        if grad_scale is not None:
            a.flat_grad.mul_(grad_scale.to(a.flat_grad.dtype).reciprocal())
            stale = True
        # no clip_grad_norm_ this iteration: the guard still needs the norm (clip stays 0)
        if stale:
            self._norm_sq()
        # GradScaler's own verdict folded into the guard on the device: if anything between unscale_ and step has sanitised
        # the gradients (nan_to_num, a custom clip), the norm is finite again while the scaler still backs its scale off -- the
        # reference would skip that step.  found_inf > 0 makes the norm non-finite: inf * 0 would be NaN, hence the where.
        found_inf = getattr(self, "found_inf", None)
        if found_inf is not None:
            self.gnorm_sq.copy_(torch.where(found_inf.to(self.gnorm_sq.device).reshape(-1)[:1] > 0, torch.full_like(self.gnorm_sq, float("inf")), self.gnorm_sq))
        if _DEV_STEP:
            # the schedule's step count = calls - drops, formed on the device (no host read of the guard's decision)
            L.check(L.lib().ft_radam_step_dev(L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.flat_m), L.ptr(self.flat_v), a.numel,
                                              L.ptr(self.gnorm_sq), clip, g["lr"], beta1, beta2, g["eps"], g["weight_decay"],
                                              self._step + self._skipped_applied, L.ptr(self._skipped), L.stream()), "ft_radam_step_dev")
        else:
            ss, rect = self.step_size_for(self._step, g["lr"], beta1, beta2)
            L.check(L.lib().ft_radam_step(L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.flat_m), L.ptr(self.flat_v),
                                          a.numel, L.ptr(self.gnorm_sq), clip, g["lr"], beta1, beta2,
                                          g["eps"], g["weight_decay"], ss, int(rect), L.ptr(self._skipped), L.stream()), "ft_radam_step")
        self._clip = 0.0
        self._have_norm = False
        for off, k, pv, mv, vv in keep:
            a.flat_param[off:off + k].copy_(pv)
            self.flat_m[off:off + k].copy_(mv)
            self.flat_v[off:off + k].copy_(vv)
        held = {off for off, _ in skipped}
        for p, off in zip(a.params, a.offsets):
            if off not in held:                  # radam.py:57-58: a parameter without a gradient keeps its step count
                self.state[p]["step"] = self._step
        return loss

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()


def poison_from_status(flat: torch.Tensor):
    """Enqueue `if (persistent-recurrence status word != 0) flat[0] = NaN` (ft_poison_if_nonzero): no host synchronisation.
    A no-op when no persistent kernel has run on this device."""
    from . import ops
    st = ops._PERSIST.get(flat.device)
    if st is not None:
        L.check(L.lib().ft_poison_if_nonzero(L.ptr(st.status), L.ptr(flat), L.stream()), "ft_poison_if_nonzero")
        ops.persist_consume_failure(flat.device)
