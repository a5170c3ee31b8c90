"""Data-parallel runtime: the host-side mirror of the reference's distributed.py (same three entry
points, same call sites in train.py:215-252, :306-316) re-designed for one process per MI355X over
RCCL / xGMI.

reference (distributed.py:96-120)                 here
-------------------------------------------------  ------------------------------------------------
per step: torch.cat of 68 grads (244 MB copy),     ONE persistent flat fp32 gradient arena; every
all_reduce, /= world, 68 copy_ back                 param.grad is a VIEW into it, so the step is a
                                                    in-place RCCL all-reduce(AVG) of the arena, one
                                                    bucket per flow (regime agreed across ranks at wrap time)
68 parameter broadcasts at start-up                 one broadcast of a flat parameter arena
4 scalar all-reduces + .item() per step             reduce_tensors(): one 4-float all-reduce

`torch.distributed` backend "nccl" IS RCCL on ROCm.  gloo is accepted so the N>1 logic is testable on CPU.
"""
from __future__ import annotations

import os
import weakref
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


_ARENAS = []          # weak references to the live arenas (one per process in practice)


def arena_slot(W):
    """(arena, parameter index) if W is exactly the storage of a parameter of a live arena (FlatArena.slot_of), else None"""
    dead = False
    for r in _ARENAS:
        a = r()
        if a is None:
            dead = True
            continue
        i = a.slot_of(W)
        if i is not None:
            return a, i
    if dead:
        _ARENAS[:] = [r for r in _ARENAS if r() is not None]
    return None


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
        # weight gradients written IN PLACE (grad_view_for_pass): parameter storage address -> index, the backward pass being served
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        self._pass, self._pass_clean, self._handed = None, False, set()
        _ARENAS.append(weakref.ref(self))

    @classmethod
    def for_params(cls, params, flatten_params: bool = True) -> "FlatArena":
        """Reuse the arena these parameters already live in (optimizer and DP wrapper share ONE arena,
        whichever is constructed first: train.py builds the optimizer at :230 and wraps the model at :251)."""
        plist = [p for p in params if p.requires_grad]
        a = getattr(plist[0], "_ft_arena", None) if plist else None
        if a is not None and [id(p) for p in a.params] == [id(p) for p in plist] and (a.flat_param is not None or not flatten_params):
            return a
        return cls(plist, flatten_params)

    def zero_grad(self, lazy: bool = True):
        """lazy (default, round 4): p.grad = None on every parameter -- what torch's own zero_grad(set_to_none=True) does (train.py:282
        `model.zero_grad()`).  The next backward then HANDS its gradient tensors over (autograd steals them: no `grad += dW` kernel
        per parameter, 68 of them per step, and no 244 MB memset), and whoever needs the flat arena next -- the optimizer, the
        all-reduce -- pulls them in with ONE multi-tensor copy (adopt_stray_grads).  lazy=False: zero the arena and keep every
        .grad a view into it (gradients accumulate in place; the round-1..3 behaviour)."""
        if lazy:
            for p in self.params:
                p.grad = None
            return
        self.flat_grad.zero_()
        self.adopt_stray_grads(copy=False)

    def slot_of(self, W):
        """index of the parameter whose storage W covers exactly (the parameter itself, or a reshape of it: a Conv1d weight viewed as
        a matrix), else None; checked against the LIVE parameter, so a stale address can never match"""
        i = self._by_ptr.get(W.data_ptr())
        if i is None:
            return None
        p = self.params[i]
        if p.data_ptr() != W.data_ptr() or p.numel() != W.numel() or not W.is_contiguous() or W.dtype != torch.float32:
            return None
        return i

    def grad_view_for_pass(self, i, task, shape):
        """The arena slice of parameter i as the ZEROED output of its weight-gradient kernel in backward pass `task` (the autograd
        engine's graph-task id), or None.  Served only in a pass that begins with every .grad None (zero_grad's default): the arena
        is then zeroed ONCE (one fill instead of the zeroed slab of the same size), the split-K GEMMs accumulate straight into it,
        autograd adopts the returned view as .grad (a fresh tensor object each time: AccumulateGrad takes it over without a copy) and
        adopt_stray_grads finds it already in place -- no 2 x 244 MB multi-tensor copy per step.  A second contribution to the same
        parameter in the pass, or any pass that starts with gradients in place (accumulation), gets None: the caller's own buffer is
        then accumulated by autograd as before (the engine adds two contributions of one pass out of place when their storage is
        shared, as an arena view's is: that sum is a stray tensor and the adoption copies it in -- tests/test_host_cpu.py)."""
        if self._pass != task:
            self._pass, self._handed = task, set()
            self._pass_clean = all(p.grad is None for p in self.params)
            if self._pass_clean:
                self.flat_grad.zero_()
        if not self._pass_clean or i in self._handed or self.params[i].grad is not None:
            return None
        self._handed.add(i)
        off = self.offsets[i]
        return self.flat_grad[off:off + self.params[i].numel()].view(shape)

    def _views(self):
        v = getattr(self, "_grad_views", None)
        if v is None:
            v = self._grad_views = [self.flat_grad[off:off + p.numel()].view_as(p.data) for p, off in zip(self.params, self.offsets)]
        return v

    def adopt_stray_grads(self, copy=True, only=None):
        """If something replaced p.grad (zero_grad(set_to_none=True) followed by backward: the usual case since round 4), pull it
        back into the arena so the single-collective / fused-optimizer path stays valid: one multi-tensor copy for all of them.
        A parameter whose grad is None got NO gradient: with copy=True its arena slice is zeroed (the slice still holds the previous
        iteration's values) and it is reported back as (offset, numel) so that the optimizer can skip it like radam.py:57-58; its
        .grad stays None."""
        skipped, dst, src = [], [], []
        views = self._views()
        self.adopt_copied = False            # did this call move gradient VALUES into the arena (as opposed to zero-filling None slices)?
        idx = range(len(self.params)) if only is None else only
        for i in idx:
            p, off = self.params[i], self.offsets[i]
            k = p.numel()
            g = p.grad
            if g is None:
                if copy:
                    views[i].zero_()
                    skipped.append((off, k))
                else:
                    p.grad = views[i]
            elif not (self._ptr_lo <= g.data_ptr() < self._ptr_hi):
                if copy:
                    if g.dtype == views[i].dtype and g.shape == views[i].shape:
                        dst.append(views[i]); src.append(g)
                    else:
                        views[i].copy_(g)
                    self.adopt_copied = True
                p.grad = views[i]
        if len(dst) == 1:
            dst[0].copy_(src[0])
        elif dst:
            torch._foreach_copy_(dst, src)
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


def _agree_min(flag: int) -> int:
    """the minimum of an integer flag over all ranks (one tiny collective at wrap time): every rank then runs the same regime"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(flag)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(flag)], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def apply_gradient_allreduce(module):
    """distributed.py:81-133: same contract (the module keeps its class; gradients are averaged across ranks once per
    backward).  The persistent flat gradient arena is reduced IN PLACE, bucket by bucket (one bucket per flow, ~111 MB, plus
    encoder + embeddings), each bucket one async RCCL all-reduce(AVG), all waited for by the end-of-backward callback.

    Two regimes, chosen ONCE at wrap time and agreed across ranks (all_reduce(MIN) of the flag), so that every rank always issues
    the same collectives in the same order -- never from per-rank run-time state:
      default            ONE in-place all-reduce(AVG) of the whole arena from the end-of-backward callback (north_star: "a single
                         RCCL all-reduce per step"; distributed.py:96-120 also issues one).  The default training path runs the
                         persistent recurrence kernels (csrc/lstm_persist.hip: all 256 CUs, one workgroup per CU, workgroups spin
                         on each other), which leave no CU for a collective's kernel to run beside them, so nothing is gained by
                         starting earlier -- and four back-to-back bucket collectives would only pay four latencies.
      FLOWTRON_DP_OVERLAP=1 (on every rank)   one bucket per flow (~111 MB) plus encoder + embeddings; a bucket is handed to RCCL
                         the moment the last of its gradients has been accumulated (post-accumulate-grad hooks), under the
                         remaining backward.  Meant for the launch-per-step kernels (FLOWTRON_LSTM_PERSIST=0, fp32 mode,
                         H != 1024), whose launch chains leave most of the chip idle.
    Launch order inside a regime is a function of the autograd graph only (completion order of the buckets, then arena order
    for whatever was not launched by a hook), identical on all ranks.
    FLOWTRON_DP_BUCKETS=flow forces the per-flow buckets without overlap (back to back at the end of backward: the round-3
    default, kept for A/B measurements), FLOWTRON_DP_BUCKETS=1 forces the single all-reduce in either regime.

    Robustness: the hook state is re-armed by the forward pre-hook AND by the first gradient hook that fires a second time within
    what the state believes to be one pass (a backward that raised -- the engine then never runs the end-of-backward callback --
    followed by another backward without a forward in between: the hooks fire in graph order, so the very first hook of the new
    pass is recognised), so stale counters can neither survive nor release a bucket early; before a bucket leaves, `if
    (persistent-recurrence status) grad = NaN` is enqueued on it (optim.poison_from_status), which makes a failed recurrence on
    ONE rank drop that optimizer step on EVERY rank (ft_radam_step's non-finite-norm guard) instead of averaging garbage into
    the weights.  Like distributed.py:111-126, one reduction per FORWARD: a second backward through the same forward
    (retain_graph) accumulates locally and is not reduced again.

    `module._comm_timing = True` (bench.py) brackets the end-of-backward exchange with HIP events on the compute stream:
    `module._comm_events` then holds one (start, end) pair per step = the communication time the step could not hide."""
    arena = FlatArena.for_params(list(module.parameters()), flatten_params=True)
    module._grad_arena = arena
    if dist.is_initialized():
        dist.broadcast(arena.flat_param, 0)                      # C1: one broadcast instead of 68
        for b in module.buffers():
            dist.broadcast(b, 0)
    module.needs_reduction = True
    overlap = bool(_agree_min(os.environ.get("FLOWTRON_DP_OVERLAP", "0") == "1"))
    bucketed = bool(_agree_min(os.environ.get("FLOWTRON_DP_BUCKETS", "flow" if overlap else "1") != "1"))
    overlap = overlap and bucketed
    buckets = gradient_buckets(module, arena) if bucketed else [("all", 0, arena.numel, list(range(len(arena.params))))]
    bucket_of = {}
    for bi, (_, _, _, idx) in enumerate(buckets):
        for i in idx:
            bucket_of[id(arena.params[i])] = bi
    state = {"left": [len(b[3]) for b in buckets], "launched": [False] * len(buckets), "pending": [], "queued": False, "seen": set()}
    module._grad_buckets = buckets
    module._grad_overlap = overlap
    module._comm_timing = False
    module._comm_events = []
    module._grad_bucket_log = []                                 # order in which buckets were launched in the last backward (tests)
    on_gpu = arena.flat_grad.is_cuda

    def rearm():
        for pend in state["pending"]:                            # collectives of a pass whose callback never ran: every rank issued
            _finish(pend)                                        # them (same graph), so waiting is safe; their result is discarded
        state["pending"] = []
        state["left"] = [len(b[3]) for b in buckets]
        state["launched"] = [False] * len(buckets)
        state["queued"] = False
        state["seen"] = set()

    def launch(bi):
        if state["launched"][bi] or not dist.is_initialized():
            return
        state["launched"][bi] = True
        _, lo, hi, idx = buckets[bi]
        arena.adopt_stray_grads(copy=True, only=idx)
        if on_gpu:
            from .optim import poison_from_status
            poison_from_status(arena.flat_grad[lo:hi])
        module._grad_bucket_log.append(buckets[bi][0])
        state["pending"].append(_avg_all_reduce(arena.flat_grad[lo:hi], async_op=True))

    def finish_backward():
        state["queued"] = False
        if module.needs_reduction:
            module.needs_reduction = False
            timed = module._comm_timing and on_gpu and dist.is_initialized()
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for bi in range(len(buckets)):                       # everything a hook did not launch, in arena order
                launch(bi)
            for pend in state["pending"]:
                _finish(pend)
            if timed:
                ev[1].record()
                module._comm_events.append(ev)
        state["pending"] = []
        state["left"] = [len(b[3]) for b in buckets]
        state["launched"] = [False] * len(buckets)
        state["seen"] = set()

    def on_grad(p):
        if not module.needs_reduction:
            return
        if id(p) in state["seen"]:
            # this parameter's hook already fired in what the state takes for the current pass: that pass never reached its
            # end-of-backward callback (it raised) and a NEW backward has begun without a forward in between -- start over
            rearm()
        state["seen"].add(id(p))
        if not state["queued"]:
            state["queued"] = True
            module._grad_bucket_log = []
            Variable._execution_engine.queue_callback(finish_backward)
        bi = bucket_of[id(p)]
        state["left"][bi] -= 1
        if overlap and state["left"][bi] == 0:
            launch(bi)

    for p in arena.params:
        p.register_post_accumulate_grad_hook(on_grad)

    def before_forward(self, input):
        rearm()

    def set_needs_reduction(self, input, output):
        self.needs_reduction = True

    module.register_forward_pre_hook(before_forward)
    module.register_forward_hook(set_needs_reduction)

    def zero_grad(set_to_none: bool = False):                    # keep grads inside the arena (train.py:282)
        arena.zero_grad()

    module.zero_grad = zero_grad
    return module
