"""N > 1 data-parallel logic on CPU (gloo, world_size 2): the flat gradient arena, the single all-reduce per backward
queued from the parameter hooks (distributed.py:96-132 contract), the start-up broadcast, and reduce_tensor(s)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _plain(res):
    """tensors cross the queue BY VALUE (numpy): torch's shared-memory handles need the sender alive at unpickling time"""
    return {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v) for k, v in res.items()}


def _tensors(res):
    import numpy as np
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in res.items()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import distributed as D
    D.init_distributed(rank, world, "gloo", None)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must equalise
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net = D.apply_gradient_allreduce(net)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.manual_seed(7 + rank)                        # different data per rank (sharded utterances)
    x = torch.randn(4, 6)
    res = {}
    for it in range(2):
        net.zero_grad()
        loss = net(x).pow(2).mean()
        loss.backward()
        res["g%d" % it] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    arena = net._grad_arena
    res["views"] = all(arena._ptr_lo <= p.grad.data_ptr() < arena._ptr_hi for p in net.parameters())
    res["w0"] = w0
    res["x"] = x
    res["rt"] = D.reduce_tensor(torch.tensor(float(rank + 1)), world)
    res["rts"] = torch.stack(D.reduce_tensors([torch.tensor(1.0 * rank), torch.tensor(2.0), torch.tensor(3.0 + rank), torch.tensor(0.5)], world))
    # a stray grad (zero_grad(set_to_none=True) semantics) must be pulled back into the arena
    for p in net.parameters():
        p.grad = None
    net.needs_reduction = True
    net(x).pow(2).mean().backward()
    res["g_stray"] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    res["views2"] = all(arena._ptr_lo <= p.grad.data_ptr() < arena._ptr_hi for p in net.parameters())
    q.put((rank, _plain(res)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {r: _tensors(v) for r, v in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = out[0], out[1]
    assert torch.equal(r0["w0"], r1["w0"])                         # C1: broadcast from rank 0
    assert torch.allclose(r0["g0"], r1["g0"]) and torch.allclose(r0["g1"], r1["g1"])
    assert torch.allclose(r0["g0"], r0["g1"])                      # zero_grad keeps grads in the arena, no accumulation
    assert r0["views"] and r1["views"] and r0["views2"] and r1["views2"]
    assert torch.allclose(r0["g_stray"], r0["g0"], atol=1e-6)
    # averaged gradient == mean of the two local gradients computed independently
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    off = 0
    for p in net.parameters():
        p.data.copy_(r0["w0"][off:off + p.numel()].view_as(p))
        off += p.numel()
    gs = []
    for r in (r0, r1):
        net.zero_grad()
        net(r["x"]).pow(2).mean().backward()
        gs.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
    assert torch.allclose(r0["g0"], (gs[0] + gs[1]) / 2, atol=1e-6)
    assert abs(r0["rt"].item() - 1.5) < 1e-6
    assert torch.allclose(r0["rts"], torch.tensor([0.5, 2.0, 3.5, 0.5]))


class _ToyFlows(torch.nn.Module):
    """the parameter layout of Flowtron as the bucketing sees it: front (embedding + encoder) then flows.0, flows.1 ..;
    forward runs flows 0 .. F-1 on the encoder output, so backward completes flows.F-1 first and the front last."""

    def __init__(self, n_flows=3, with_unused=False):
        super().__init__()
        self.embedding = torch.nn.Embedding(11, 6)
        self.encoder = torch.nn.Linear(6, 6)
        self.flows = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(12, 8), torch.nn.Tanh(), torch.nn.Linear(8, 6))
                                          for _ in range(n_flows)])
        if with_unused:
            self.flows[1].register_parameter("unused", torch.nn.Parameter(torch.ones(5)))

    def forward(self, ids, x):
        enc = self.encoder(self.embedding(ids)).mean(1)
        for f in self.flows:
            x = x + f(torch.cat([x, enc], 1))
        return x


def _bucket_worker(rank, world, port, q, mode, with_unused, overlap="1", raise_in_backward=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode is None:
        os.environ.pop("FLOWTRON_DP_BUCKETS", None)              # the default: one all-reduce, per-flow buckets only with overlap
    else:
        os.environ["FLOWTRON_DP_BUCKETS"] = mode
    os.environ["FLOWTRON_DP_OVERLAP"] = overlap[rank] if isinstance(overlap, (list, tuple)) else overlap
    import distributed as D
    D.init_distributed(rank, world, "gloo", None)
    torch.manual_seed(5)
    net = D.apply_gradient_allreduce(_ToyFlows(3, with_unused))
    torch.manual_seed(50 + rank)
    ids, x = torch.randint(0, 11, (4, 3)), torch.randn(4, 6)
    res = {"ids": ids, "x": x, "overlap": bool(net._grad_overlap)}
    if raise_in_backward:
        # a backward pass that dies half way (OOM, a raising hook): the autograd engine never runs the end-of-backward callback, so
        # `queued` / `left` / `pending` are stale; the forward pre-hook must re-arm them or every later step trains on unreduced
        # gradients.  Every rank raises at the same node (same graph), so the collectives already issued match.
        class _Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, v):
                return v.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")
        net.zero_grad()
        h = net.flows[0]
        y = net(ids, x)
        hooked = _Boom.apply(net.flows[1][0].weight)             # flows.2 has completed when this node is reached
        try:
            (y.pow(2).mean() + 0.0 * hooked.sum()).backward(retain_graph=raise_in_backward == "retry")
            res["raised"] = False
        except RuntimeError:
            res["raised"] = True
        if raise_in_backward == "retry":
            # ... and the caller retries the backward pass WITHOUT a new forward (no pre-hook re-arms anything): the first gradient
            # hook that fires a second time must be taken for the start of a new pass
            net._grad_arena.flat_grad.zero_()
            y.pow(2).mean().backward()
            res["g_retry"] = net._grad_arena.flat_grad.clone()
            res["log_retry"] = list(net._grad_bucket_log)
    for it in range(2):
        net.zero_grad()
        net(ids, x).pow(2).mean().backward()
        res["g%d" % it] = net._grad_arena.flat_grad.clone()
        res["log%d" % it] = list(net._grad_bucket_log)
    res["buckets"] = [(n, lo, hi) for n, lo, hi, _ in net._grad_buckets]
    res["unused_grad_is_none"] = (not with_unused) or net.flows[1].unused.grad is None or float(net.flows[1].unused.grad.abs().sum()) == 0.0
    q.put((rank, _plain(res)))
    dist.barrier()
    dist.destroy_process_group()


def _run_bucket_world(mode, with_unused, overlap="1", raise_in_backward=False):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q, mode, with_unused, overlap, raise_in_backward)) for r in range(world)]
    for p in procs:
        p.start()
    out = {r: _tensors(v) for r, v in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("with_unused", [False, True])
def test_per_flow_buckets_overlap_order_and_equal_the_single_allreduce(with_unused):
    """FLOWTRON_DP_OVERLAP=1 on every rank: per-flow buckets launched in the order backward completes them (last flow first, front last), every rank ends with
    the same averaged arena, identical to FLOWTRON_DP_BUCKETS=1 (one all-reduce at the end) and to the mean of the local
    gradients; a bucket holding a parameter that never gets a gradient is swept up by the end-of-backward callback."""
    a = _run_bucket_world("flow", with_unused)
    b = _run_bucket_world("1", with_unused)
    names = [n for n, _, _ in a[0]["buckets"]]
    assert names == ["embedding+encoder", "flows.0", "flows.1", "flows.2"]
    los = [lo for _, lo, _ in a[0]["buckets"]]
    his = [hi for _, _, hi in a[0]["buckets"]]
    assert los[0] == 0 and los[1:] == his[:-1] and his[-1] == a[0]["g0"].numel()        # contiguous cover of the arena
    for it in (0, 1):
        if with_unused:     # flows.1 never completes by itself: launched by the final callback, after the front
            assert a[0]["log%d" % it] == ["flows.2", "flows.0", "embedding+encoder", "flows.1"]
        else:
            assert a[0]["log%d" % it] == ["flows.2", "flows.1", "flows.0", "embedding+encoder"]
        assert b[0]["log%d" % it] == ["all"]
        assert torch.equal(a[0]["g%d" % it], a[1]["g%d" % it])
        assert torch.allclose(a[0]["g%d" % it], b[0]["g%d" % it], atol=1e-7)
    assert a[0]["unused_grad_is_none"]
    # against the mean of the two local gradients
    torch.manual_seed(5)
    net = _ToyFlows(3, with_unused)
    gs = []
    for r in (0, 1):
        net.zero_grad()
        net(a[r]["ids"], a[r]["x"]).pow(2).mean().backward()
        gs.append([p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in net.parameters()])
    import distributed as D
    arena = D.FlatArena(list(net.parameters()))
    for p, off, g0, g1 in zip(arena.params, arena.offsets, gs[0], gs[1]):
        assert torch.allclose(a[0]["g0"][off:off + p.numel()].view_as(p), (g0 + g1) / 2, atol=1e-6)



def test_default_regime_is_end_of_backward_in_arena_order_and_ranks_agree_on_it():
    """Default (no FLOWTRON_DP_OVERLAP): the buckets leave from the end-of-backward callback in arena order.  A rank whose
    environment asks for overlap while another's does not must NOT issue a different collective sequence: the regime is the
    MINIMUM over ranks, agreed once at wrap time -- both ranks run the default regime and end with the same averaged arena."""
    a = _run_bucket_world("flow", False, overlap="0")
    mixed = _run_bucket_world("flow", False, overlap=["1", "0"])
    ref = _run_bucket_world("flow", False, overlap="1")
    order = ["embedding+encoder", "flows.0", "flows.1", "flows.2"]
    for world in (a, mixed):
        for r in (0, 1):
            assert world[r]["overlap"] is False
            assert world[r]["log0"] == order and world[r]["log1"] == order
        assert torch.equal(world[0]["g0"], world[1]["g0"]) and torch.equal(world[0]["g1"], world[1]["g1"])
        assert torch.allclose(world[0]["g0"], ref[0]["g0"], atol=1e-7)
    assert ref[0]["overlap"] is True and ref[1]["overlap"] is True


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_hook_state_is_rearmed_after_a_backward_pass_that_raised(overlap):
    """the end-of-backward callback does not run when backward raises; the next forward must re-arm the hook state
    (and wait for collectives the dead pass had already issued) so that the following steps still reduce every bucket."""
    a = _run_bucket_world("flow", False, overlap=overlap, raise_in_backward=True)
    ref = _run_bucket_world("flow", False, overlap=overlap)
    for r in (0, 1):
        assert a[r]["raised"] is True
        for it in (0, 1):
            assert len(a[r]["log%d" % it]) == 4, a[r]["log%d" % it]
            assert torch.allclose(a[r]["g%d" % it], ref[r]["g%d" % it], atol=1e-7)
    assert torch.equal(a[0]["g1"], a[1]["g1"])


def test_default_is_one_allreduce_of_the_arena_and_overlap_selects_the_per_flow_buckets():
    """VERDICT r3 #7 / north_star "a single RCCL all-reduce per step": with no FLOWTRON_DP_BUCKETS in the environment the
    end-of-backward exchange is ONE collective over the whole arena; FLOWTRON_DP_OVERLAP=1 (on every rank) switches to the per-flow
    buckets launched under the remaining backward.  Same averaged gradients either way."""
    one = _run_bucket_world(None, False, overlap="0")
    ovl = _run_bucket_world(None, False, overlap="1")
    for r in (0, 1):
        assert one[r]["overlap"] is False and ovl[r]["overlap"] is True
        for it in (0, 1):
            assert one[r]["log%d" % it] == ["all"]
            assert ovl[r]["log%d" % it] == ["flows.2", "flows.1", "flows.0", "embedding+encoder"]
    assert [n for n, _, _ in one[0]["buckets"]] == ["all"]
    assert torch.equal(one[0]["g1"], one[1]["g1"]) and torch.allclose(one[0]["g1"], ovl[0]["g1"], atol=1e-7)


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_a_backward_retried_without_a_new_forward_reduces_every_bucket_once(overlap):
    """a backward pass that raised leaves partly counted buckets behind; when the caller retries the backward through the
    retained graph (no forward, so no pre-hook), the hook state starts over at the first hook that fires twice -- no bucket leaves
    early with half of its gradients, none is left out, and nothing raises."""
    a = _run_bucket_world("flow", False, overlap=overlap, raise_in_backward="retry")
    ref = _run_bucket_world("flow", False, overlap=overlap)
    for r in (0, 1):
        assert a[r]["raised"] is True
        assert sorted(a[r]["log_retry"]) == ["embedding+encoder", "flows.0", "flows.1", "flows.2"]
        assert torch.allclose(a[r]["g_retry"], ref[r]["g0"], atol=1e-7)
    assert torch.equal(a[0]["g_retry"], a[1]["g_retry"])
