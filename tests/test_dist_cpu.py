"""N > 1 data-parallel logic on CPU (gloo, world_size 2): the flat gradient arena, the single all-reduce per backward
queued from the parameter hooks (distributed.py:96-132 contract), the start-up broadcast, and reduce_tensor(s)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
    q.put((rank, res))
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
    out = dict(q.get(timeout=120) for _ in range(world))
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
