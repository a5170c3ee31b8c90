"""Data-parallel runtime: the host-side mirror of the reference's distributed.py (same three entry
points, same call sites in train.py:215-252, :306-316) re-designed for one process per MI355X over
RCCL / xGMI.

reference (distributed.py:96-120)                 here
-------------------------------------------------  ------------------------------------------------
per step: torch.cat of 68 grads (244 MB copy),     ONE persistent flat fp32 gradient arena; every
all_reduce, /= world, 68 copy_ back                 param.grad is a VIEW into it, so the step is a
                                                    in-place RCCL all-reduce(AVG) of the arena, one
                                                    bucket per flow, launched as its gradients complete
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


def _avg_op():
    """RCCL divides inside the reduction kernel (ReduceOp.AVG): no separate 244 MB `flat /= world_size` pass.  gloo (the CPU
    test backend) has no AVG: SUM, then divide."""
    return dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM


def _avg_all_reduce(flat: torch.Tensor, async_op: bool = False):
    """in-place average of `flat` over the ranks; async_op=True returns a waitable (call _finish on it)."""
    op = _avg_op()
    work = dist.all_reduce(flat, op=op, async_op=async_op)
    if async_op:
        return (work, flat, op)
    if op == dist.ReduceOp.SUM:
        flat /= dist.get_world_size()
    return None


def _finish(pending):
    work, flat, op = pending
    work.wait()                                               # the compute stream now waits for the collective's stream
    if op == dist.ReduceOp.SUM:
        flat /= dist.get_world_size()


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

    def adopt_stray_grads(self, copy=True, only=None):
        """If something replaced p.grad (e.g. torch's default zero_grad(set_to_none=True) followed by backward), pull it
        back into the arena so the single-collective path stays valid.  A parameter whose grad is None got NO gradient:
        with copy=True its arena slice is zeroed (the slice still holds the previous iteration's values) and it is
        reported back as (offset, numel) so that the optimizer can skip it like radam.py:57-58; its .grad stays None."""
        skipped = []
        items = zip(self.params, self.offsets) if only is None else ((self.params[i], self.offsets[i]) for i in only)
        for p, off in items:
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


def gradient_buckets(module, arena: "FlatArena"):
    """Contiguous arena ranges that become final together during backward: one per flow (`flows.<i>.*`) and one per run of
    adjacent non-flow parameters (embeddings, encoder: their gradients complete last because every flow reads the encoder
    output), named after the top-level modules it covers.  Returns [(name, lo, hi, [param indices])] in arena order."""
    names = {id(p): n for n, p in module.named_parameters()}
    groups = []
    for i, p in enumerate(arena.params):
        n = names.get(id(p), "")
        flow = n.startswith("flows.")
        key = ".".join(n.split(".")[:2]) if flow else n.split(".")[0]
        if groups and (groups[-1][0] == key if flow else not groups[-1][2]):
            groups[-1][1].append(i)
            if not flow and key not in groups[-1][0].split("+"):
                groups[-1][0] += "+" + key
        else:
            groups.append([key, [i], flow])
    out = []
    for key, idx, _ in groups:
        lo = arena.offsets[idx[0]]
        hi = arena.offsets[idx[-1] + 1] if idx[-1] + 1 < len(arena.offsets) else arena.numel
        out.append((key, lo, hi, idx))
    return out


def apply_gradient_allreduce(module):
    """distributed.py:81-133: same contract (the module keeps its class; gradients are averaged across ranks once per
    backward).  The persistent flat gradient arena is reduced IN PLACE, bucket by bucket: a bucket (one flow's parameters,
    ~120 MB) is handed to RCCL the moment the last of its gradients has been accumulated (post-accumulate-grad hooks), so
    the all-reduce of flow F-1 runs on RCCL's stream under the backward recurrences of flows F-2 .. 0 -- launch chains that
    leave most of the chip idle -- and only the last bucket (embeddings + encoder, 9 MB) is exposed.  Whatever was not
    launched by then (parameters without a gradient) is reduced by the end-of-backward callback, which also waits.
    FLOWTRON_DP_BUCKETS=1 falls back to ONE all-reduce of the whole arena at the end of backward.

    Co-residency rule.  The persistent recurrence kernels (csrc/lstm_persist.hip) need all 256 CUs at once, one workgroup
    per CU owning the whole register file, and spin on each other: an RCCL kernel that is only partly resident beside a partly
    resident persistent grid on two ranks can wait on each other across ranks (ring channel c needs channel c resident on
    every rank) until the persistent kernel's 0.5 s timeout fires.  So when the forward pass of this step used them, the
    buckets are NOT launched under backward: they go out back to back from the end-of-backward callback, when nothing else
    is queued on the device (still in place, still AVG-folded, still pipelined bucket after bucket).  The overlap applies to
    the launch-per-step kernels (FLOWTRON_LSTM_PERSIST=0, batches > 32, H != 1024).  FLOWTRON_DP_OVERLAP=1 | 0 overrides."""
    ws = dist.get_world_size() if dist.is_initialized() else 1
    arena = FlatArena.for_params(list(module.parameters()), flatten_params=True)
    module._grad_arena = arena
    if dist.is_initialized():
        dist.broadcast(arena.flat_param, 0)                      # C1: one broadcast instead of 68
        for b in module.buffers():
            dist.broadcast(b, 0)
    module.needs_reduction = True
    bucketed = os.environ.get("FLOWTRON_DP_BUCKETS", "flow") != "1"
    buckets = gradient_buckets(module, arena) if bucketed else [("all", 0, arena.numel, list(range(len(arena.params))))]
    bucket_of = {}
    for bi, (_, _, _, idx) in enumerate(buckets):
        for i in idx:
            bucket_of[id(arena.params[i])] = bi
    state = {"left": [len(b[3]) for b in buckets], "launched": [False] * len(buckets), "pending": [], "queued": False,
             "persist_before": 0, "overlap": True}
    force = os.environ.get("FLOWTRON_DP_OVERLAP", "auto")

    def persist_launches():
        try:
            from . import ops
            return ops.PERSIST_LAUNCHES
        except Exception:                                        # CPU-only host (gloo tests): no HIP library, no persistent kernels
            return 0
    module._grad_buckets = buckets
    module._grad_bucket_log = []                                 # order in which buckets were launched in the last backward (tests)

    def launch(bi):
        if state["launched"][bi] or not dist.is_initialized():
            return
        state["launched"][bi] = True
        _, lo, hi, idx = buckets[bi]
        arena.adopt_stray_grads(copy=True, only=idx)
        module._grad_bucket_log.append(buckets[bi][0])
        state["pending"].append(_avg_all_reduce(arena.flat_grad[lo:hi], async_op=True))

    def finish_backward():
        state["queued"] = False
        if module.needs_reduction:
            module.needs_reduction = False
            for bi in range(len(buckets)):                       # leftovers: buckets with a gradient-less parameter
                launch(bi)
            for pend in state["pending"]:
                _finish(pend)
        state["pending"] = []
        state["left"] = [len(b[3]) for b in buckets]
        state["launched"] = [False] * len(buckets)

    def on_grad(p):
        if not module.needs_reduction:
            return
        if not state["queued"]:
            state["queued"] = True
            module._grad_bucket_log = []
            Variable._execution_engine.queue_callback(finish_backward)
        bi = bucket_of[id(p)]
        state["left"][bi] -= 1
        if bucketed and state["overlap"] and state["left"][bi] == 0:
            launch(bi)

    for p in arena.params:
        p.register_post_accumulate_grad_hook(on_grad)

    def before_forward(self, input):
        state["persist_before"] = persist_launches()

    def set_needs_reduction(self, input, output):
        self.needs_reduction = True
        used_persistent = persist_launches() != state["persist_before"]
        state["overlap"] = (force == "1") or (force != "0" and not used_persistent)

    module.register_forward_pre_hook(before_forward)
    module.register_forward_hook(set_needs_reduction)

    def zero_grad(set_to_none: bool = False):                    # keep grads inside the arena (train.py:282)
        arena.zero_grad()

    module.zero_grad = zero_grad
    return module
