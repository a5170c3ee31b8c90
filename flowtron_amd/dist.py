"""Data-parallel runtime: the host-side mirror of the reference's distributed.py (same three entry
points, same call sites in train.py:215-252, :306-316) re-designed for one process per MI355X over
RCCL / xGMI.

reference (distributed.py:96-120)                 here
-------------------------------------------------  ------------------------------------------------
per step: torch.cat of 68 grads (244 MB copy),     ONE persistent flat fp32 gradient arena; every
all_reduce, /= world, 68 copy_ back                 param.grad is a VIEW into it, so the step is a
                                                    single in-place RCCL all-reduce(AVG) of the arena
68 parameter broadcasts at start-up                 one broadcast of a flat parameter arena
4 scalar all-reduces + .item() per step             reduce_tensors(): one 4-float all-reduce

`torch.distributed` backend "nccl" IS RCCL on ROCm.  gloo is accepted so the N>1 logic is testable on CPU.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist
from torch.autograd import Variable


def init_distributed(rank, num_gpus, dist_backend="nccl", dist_url=None):
    """distributed.py:28-44.  The reference ignores dist_backend/dist_url (hard-codes nccl + env
    rendez-vous); we honour dist_backend only to allow 'gloo' on CPU-only hosts (tests)."""
    if dist.is_initialized():
        return
    use_cuda = torch.cuda.is_available()
    backend = "nccl" if use_cuda else "gloo"
    if dist_backend == "gloo":
        backend = "gloo"
    assert use_cuda or backend == "gloo", "Distributed mode requires a GPU (or the gloo test backend)."
    print("> initializing distributed for rank {} out of {}".format(rank, num_gpus))
    if use_cuda:
        local = int(os.environ.get("LOCAL_RANK", rank % torch.cuda.device_count()))
        torch.cuda.set_device(local)
    master_ip = os.getenv("MASTER_ADDR", "127.0.0.1")
    master_port = os.getenv("MASTER_PORT", "6000")
    if os.getenv("TORCHELASTIC_RUN_ID") is not None or os.getenv("TORCHELASTIC_USE_AGENT_STORE") is not None:
        # launched by torch.distributed.run: its agent already serves the store on MASTER_PORT, so rank 0 must NOT
        # open a second TCP store there (tcp:// would); env:// joins the agent's store.
        dist.init_process_group(backend=backend, world_size=num_gpus, rank=rank, init_method="env://")
    else:
        dist.init_process_group(backend=backend, world_size=num_gpus, rank=rank,
                                init_method="tcp://" + master_ip + ":" + master_port)


def _avg_all_reduce(flat: torch.Tensor):
    ws = dist.get_world_size()
    if ws == 1:
        return
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)              # one in-place ring/mesh reduction of the whole arena
    flat /= ws


def reduce_tensor(tensor, num_gpus):
    """distributed.py:22-26."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def reduce_tensors(tensors: Iterable[torch.Tensor], num_gpus: int) -> List[torch.Tensor]:
    """The four per-step scalar reductions of train.py:306-316 as ONE collective."""
    tensors = list(tensors)
    flat = torch.stack([t.detach().reshape(()).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= num_gpus
    return list(flat.unbind(0))


class FlatArena:
    """Flat fp32 storage for a list of parameters: `params` (optional) and `grads` as views."""

    def __init__(self, params: List[torch.nn.Parameter], flatten_params: bool = True):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, torch.float32
        # every tensor starts on a 256-byte boundary inside the arena: the HIP kernels read weights with 16-byte
        # vector loads (W_hh rows, GEMV rows), and a [1]-element bias must not misalign what follows it.  The gaps
        # are zero in params, grads and moments, so reductions/updates over the whole arena are unaffected.
        ALIGN = 64
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.n_params = sum(p.numel() for p in self.params)
        self.numel = off                                   # arena length incl. alignment gaps
        self.flat_grad = torch.zeros(off, device=dev, dtype=dt)
        self.flat_param = torch.zeros_like(self.flat_grad) if flatten_params else None
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            if flatten_params:
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p.data)
            p.grad = self.flat_grad[off:off + k].view_as(p.data)
        self._ptr_lo = self.flat_grad.data_ptr()
        self._ptr_hi = self._ptr_lo + self.flat_grad.numel() * 4
        for p in self.params:
            p._ft_arena = self

    @classmethod
    def for_params(cls, params, flatten_params: bool = True) -> "FlatArena":
        """Reuse the arena these parameters already live in (optimizer and DP wrapper share ONE arena,
        whichever is constructed first: train.py builds the optimizer at :230 and wraps the model at :251)."""
        plist = [p for p in params if p.requires_grad]
        a = getattr(plist[0], "_ft_arena", None) if plist else None
        if a is not None and [id(p) for p in a.params] == [id(p) for p in plist] and (a.flat_param is not None or not flatten_params):
            return a
        return cls(plist, flatten_params)

    def zero_grad(self):
        self.flat_grad.zero_()
        self.adopt_stray_grads(copy=False)

    def adopt_stray_grads(self, copy=True):
        """If something replaced p.grad (e.g. torch's default zero_grad(set_to_none=True) followed by backward), pull it
        back into the arena so the single-collective path stays valid.  A parameter whose grad is None got NO gradient:
        with copy=True its arena slice is zeroed (the slice still holds the previous iteration's values) and it is
        reported back as (offset, numel) so that the optimizer can skip it like radam.py:57-58; its .grad stays None."""
        skipped = []
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            g = p.grad
            if g is None:
                if copy:
                    self.flat_grad[off:off + k].zero_()
                    skipped.append((off, k))
                else:
                    p.grad = self.flat_grad[off:off + k].view_as(p.data)
            elif not (self._ptr_lo <= g.data_ptr() < self._ptr_hi):
                view = self.flat_grad[off:off + k].view_as(p.data)
                if copy:
                    view.copy_(g)
                p.grad = view
        return skipped


def apply_gradient_allreduce(module):
    """distributed.py:81-133: same contract (the module keeps its class; gradients are averaged across ranks
    once per backward), implemented as ONE in-place all-reduce of a persistent flat gradient arena."""
    ws = dist.get_world_size() if dist.is_initialized() else 1
    arena = FlatArena.for_params(list(module.parameters()), flatten_params=True)
    module._grad_arena = arena
    if ws > 1:
        dist.broadcast(arena.flat_param, 0)                      # C1: one broadcast instead of 68
        for b in module.buffers():
            dist.broadcast(b, 0)
    module.needs_reduction = True

    def allreduce_params():
        if module.needs_reduction:
            module.needs_reduction = False
            arena.adopt_stray_grads(copy=True)
            _avg_all_reduce(arena.flat_grad)                     # C2: the only per-step collective

    def allreduce_hook(*unused):
        Variable._execution_engine.queue_callback(allreduce_params)

    for p in arena.params:
        p.register_hook(allreduce_hook)

    def set_needs_reduction(self, input, output):
        self.needs_reduction = True

    module.register_forward_hook(set_needs_reduction)

    def zero_grad(set_to_none: bool = False):                    # keep grads inside the arena (train.py:282)
        arena.zero_grad()

    module.zero_grad = zero_grad
    return module
