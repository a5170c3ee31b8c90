"""Fused clip + RAdam over the flat arenas (-m gpu): ft_sumsq + ft_radam_step against the REAL reference optimizer
(/root/reference/radam.py, golden trajectory tests/golden/radam_traj.pt made by tests/golden/make_golden_r2.py), checkpoint
resume (train.py:110-139), and the `p.grad is None` skip of radam.py:57-58."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _golden():
    return torch.load(os.path.join(HERE, "golden", "radam_traj.pt"), weights_only=False)


def _params(shapes):
    from make_golden_r2 import radam_params
    return [torch.nn.Parameter(p.clone().cuda()) for p in radam_params(shapes)]


def _set_grads(ps, step, shapes):
    from make_golden_r2 import radam_grads
    for p, g in zip(ps, radam_grads(step, shapes)):
        p.grad.copy_(g.cuda())            # grads are views into the flat arena


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_trajectory_matches_reference_radam(case):
    """12 steps (crossing the N_sma >= 5 switch at step 6), with / without global-norm clip 1.0, weight decay 1e-6 / 1e-2:
    parameters after EVERY step and the final moments against the reference optimizer, fp32 round-off tolerance."""
    from flowtron_amd.optim import RAdam
    g = _golden()
    c = g["cases"][case]
    ps = _params(g["shapes"])
    opt = RAdam(ps, lr=1e-3, weight_decay=c["wd"])
    for step, want in enumerate(c["traj"], start=1):
        _set_grads(ps, step, g["shapes"])
        if c["clip"]:
            opt.clip_grad_norm_(c["clip"])
        opt.step()
        for p, w in zip(ps, want):
            assert (p.detach().cpu() - w).abs().max().item() < 2e-6 * max(1.0, w.abs().max().item()), (case, step)
    for p, m, v in zip(ps, c["exp_avg"], c["exp_avg_sq"]):
        st = opt.state[p]
        assert st["step"] == len(c["traj"])
        assert (st["exp_avg"].cpu() - m).abs().max().item() < 1e-6 * max(1.0, m.abs().max().item())
        assert (st["exp_avg_sq"].cpu() - v).abs().max().item() < 1e-6 * max(1.0, v.abs().max().item())


def test_torch_clip_then_step_equals_fused_clip():
    """train.py:323-331 order (torch clip_grad_norm_ on p.grad, then optimizer.step()) gives the fused-clip trajectory."""
    from flowtron_amd.optim import RAdam
    g = _golden()
    c = g["cases"][2]                     # clip 1.0, wd 1e-6
    ps = _params(g["shapes"])
    opt = RAdam(ps, lr=1e-3, weight_decay=c["wd"])
    for step, want in enumerate(c["traj"][:7], start=1):
        _set_grads(ps, step, g["shapes"])
        torch.nn.utils.clip_grad_norm_(ps, c["clip"])
        opt.step()
        for p, w in zip(ps, want):
            assert (p.detach().cpu() - w).abs().max().item() < 2e-6 * max(1.0, w.abs().max().item()), step


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """3 steps -> state_dict -> torch.save/load -> fresh optimizer.load_state_dict -> 2 steps == 5 uninterrupted steps."""
    from flowtron_amd.optim import RAdam
    g = _golden()
    shapes = g["shapes"]
    a = _params(shapes)
    oa = RAdam(a, lr=1e-3, weight_decay=1e-6)
    for step in range(1, 6):
        _set_grads(a, step, shapes)
        oa.clip_grad_norm_(1.0)
        oa.step()
    b = _params(shapes)
    ob = RAdam(b, lr=1e-3, weight_decay=1e-6)
    for step in range(1, 4):
        _set_grads(b, step, shapes)
        ob.clip_grad_norm_(1.0)
        ob.step()
    torch.save({"optimizer": ob.state_dict(), "params": [p.detach().cpu() for p in b]}, tmp_path / "ck.pt")
    ck = torch.load(tmp_path / "ck.pt", map_location="cpu", weights_only=False)
    c = [torch.nn.Parameter(p.clone().cuda()) for p in ck["params"]]
    oc = RAdam(c, lr=1e-3, weight_decay=1e-6)
    oc.load_state_dict(ck["optimizer"])
    assert oc._step == 3 and float(oc.flat_m.abs().sum()) > 0
    for p in c:                                           # state tensors are views into the arenas again
        lo, hi = oc.flat_m.data_ptr(), oc.flat_m.data_ptr() + oc.flat_m.numel() * 4
        assert lo <= oc.state[p]["exp_avg"].data_ptr() < hi
    for step in range(4, 6):
        _set_grads(c, step, shapes)
        oc.clip_grad_norm_(1.0)
        oc.step()
    for pa, pc in zip(a, c):
        assert torch.equal(pa.detach(), pc.detach())
    assert torch.equal(oa.flat_m, oc.flat_m) and torch.equal(oa.flat_v, oc.flat_v)
    sd = oc.state_dict()                                  # and what is saved next is current, not the stale loaded tensors
    assert all(s["step"] == 5 for s in sd["state"].values())


def test_parameter_without_gradient_is_left_untouched():
    """radam.py:57-58 `if p.grad is None: continue` -- no decay, no moment update; the stale arena slice does not enter
    the global norm either (torch's default zero_grad(set_to_none=True) leaves None on parameters that got no gradient)."""
    from flowtron_amd.optim import RAdam
    g = _golden()
    shapes = g["shapes"]
    ps = _params(shapes)
    opt = RAdam(ps, lr=1e-3, weight_decay=1e-2)
    _set_grads(ps, 3, shapes)
    opt.clip_grad_norm_(1.0)
    opt.step()
    before = [p.detach().clone() for p in ps]
    m1 = opt.state[ps[1]]["exp_avg"].clone()
    for p in ps:
        p.grad = None                                     # model.zero_grad() of torch >= 2.0
    from make_golden_r2 import radam_grads
    new = radam_grads(4, shapes)
    for i in (0, 2, 3):
        ps[i].grad = new[i].cuda()                        # fresh (non-arena) tensors, like autograd creates; ps[1] gets none
    nsq = opt.clip_grad_norm_(1e9)
    want = sum(float((new[i] ** 2).sum()) for i in (0, 2, 3))
    assert abs(float(nsq) - want) < 1e-4 * want
    opt.step()
    assert torch.equal(ps[1].detach(), before[1]) and ps[1].grad is None
    assert torch.equal(opt.state[ps[1]]["exp_avg"], m1)
    assert not torch.equal(ps[0].detach(), before[0])


def test_non_finite_gradient_norm_drops_the_update_on_device():
    """ft_radam_step's guard: a NaN / Inf global gradient norm leaves parameters and moments untouched and counts the skip
    (device side, what GradScaler.step does for an fp16 overflow -- train.py:330); the following finite step is a normal one."""
    from flowtron_amd.optim import RAdam
    g = _golden()
    ps = _params(g["shapes"])
    opt = RAdam(ps, lr=1e-3, weight_decay=1e-6)
    _set_grads(ps, 1, g["shapes"])
    opt.clip_grad_norm_(1.0)
    opt.step()
    snap = (opt.arena.flat_param.clone(), opt.flat_m.clone(), opt.flat_v.clone())
    for bad in (float("nan"), float("inf")):
        _set_grads(ps, 2, g["shapes"])
        ps[1].grad.view(-1)[3] = bad
        opt.clip_grad_norm_(1.0)
        opt.step()
        assert torch.equal(opt.arena.flat_param, snap[0]) and torch.equal(opt.flat_m, snap[1]) and torch.equal(opt.flat_v, snap[2])
    assert opt.skipped_steps == 2
    _set_grads(ps, 2, g["shapes"])
    opt.step()                                        # no clip this time: the guard computes the norm itself
    assert not torch.equal(opt.arena.flat_param, snap[0]) and torch.isfinite(opt.arena.flat_param).all()
    assert opt.skipped_steps == 2


def test_failed_persistent_recurrence_drops_the_step_and_falls_back_without_killing_the_run():
    """A persistent recurrence that reports a time-out (status word != 0; here injected between forward and backward of a
    full-width bf16 step, which also makes the backward launches abort at their first wait) must not reach the weights and must
    not raise: the step is poisoned (ft_poison_if_nonzero) and dropped by the optimizer's guard, the host notices one launch late,
    warns ONCE, switches the device to the launch-per-step kernels, and training continues on them."""
    import warnings
    import flowtron
    from flowtron_amd import ops
    from flowtron_amd.optim import RAdam
    from oracle import synth
    os.environ["FLOWTRON_MFMA"] = "bf16"
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        cfg = dict(synth.DEFAULT_MODEL_CONFIG)
        if not ops.lstm_persist_groups(4, cfg["n_hidden"], False, 1, dev):
            pytest.skip("persistent recurrences not usable on this device")
        m = flowtron.Flowtron(**cfg)
        m.load_state_dict(synth.make_state_dict(cfg, seed=3))
        m = m.cuda().eval()
        opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
        crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth.make_batch(cfg, [40, 33, 21, 12], [12, 9, 7, 5], seed=4, with_prior=True).items()}

        def step(inject):
            opt.zero_grad()
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            if inject:
                ops.persist_status(dev).fill_(1)
            (nll + gl + 0.01 * ctc).backward()
            opt.clip_grad_norm_(1.0)
            opt.step()
            torch.cuda.synchronize()
        step(False)
        w1 = opt.arena.flat_param.clone()
        launches = ops.PERSIST_LAUNCHES
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            step(True)                                 # poisoned: dropped
            assert torch.equal(opt.arena.flat_param, w1)
            step(False)                                # host has noticed by now (or notices here): fallback kernels; at most one more drop
            step(False)
            step(False)
        msgs = [str(w.message) for w in caught if "persistent recurrence" in str(w.message)]
        assert len(msgs) == 1, msgs
        assert not ops.persist_usable(dev)
        assert 1 <= opt.skipped_steps <= 2
        assert torch.isfinite(opt.arena.flat_param).all() and not torch.equal(opt.arena.flat_param, w1)
        assert int(ops.persist_status(dev).item()) == 0
        n_after = ops.PERSIST_LAUNCHES
        step(False)
        assert ops.PERSIST_LAUNCHES == n_after          # launch-per-step kernels from now on
        assert n_after > launches                       # (the injected step still launched persistent kernels)
    finally:
        os.environ["FLOWTRON_MFMA"] = "f32"
        ops._PERSIST.clear()                            # later tests re-run the self-test and get the persistent path back
