"""Data-parallel path on the MI355X over RCCL (-m gpu): the drop-in `distributed.py` entry points on a real Flowtron module.
  * world_size 1 (always runs on the 1-GPU box): init_process_group("nccl"), start-up broadcast, per-flow buckets handed to
    RCCL from the post-accumulate-grad hooks with ReduceOp.AVG and async handles, stream wait at the end of backward -- the
    gradients must equal those of the unwrapped module, the buckets must be launched last-flow-first, and an optimizer step
    on the shared arena must work.
  * world_size 2 (self-skips unless two GPUs are visible): two ranks, different utterances; both end with the same averaged
    arena, equal to the mean of the two local gradients.
Every world runs in spawned processes so the pytest process never owns a process group."""
import math
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SMALL = dict(n_text=64, n_text_dim=128, n_speaker_dim=32, n_attn_channels=64, n_hidden=128)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_grads(cfg, sd, batch, dev):
    import flowtron
    m = flowtron.Flowtron(**cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()               # eval() = the encoder's dropout off (gradients still flow): ranks / runs comparable
    crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
    nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
    (nll + gl + 0.01 * ctc).backward()
    return {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}


def _worker(rank, world, port, q, mode, full_width=False, overlap="0", backend="nccl", shared_gpu=False, buckets=None):
    try:
        for p_ in (ROOT, HERE):
            if p_ not in sys.path:
                sys.path.insert(0, p_)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLOWTRON_MFMA=mode,
                          LOCAL_RANK="0" if shared_gpu else str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", FLOWTRON_DP_OVERLAP=overlap)
        if buckets is None:
            os.environ.pop("FLOWTRON_DP_BUCKETS", None)
        else:
            os.environ["FLOWTRON_DP_BUCKETS"] = buckets
        import distributed as D
        import flowtron
        from flowtron_amd.optim import RAdam
        from oracle import synth
        D.init_distributed(rank, world, backend, None)
        dev = torch.device("cuda", 0 if shared_gpu else rank)
        cfg = dict(synth.DEFAULT_MODEL_CONFIG)
        if not full_width:
            cfg.update(SMALL)
        cfg["n_flows"] = 2
        sd = synth.make_state_dict(cfg, seed=3)
        lens = ([40, 33, 21], [12, 9, 7]) if rank == 0 else ([37, 30, 25], [11, 10, 6])
        batch = synth.make_batch(cfg, lens[0], lens[1], seed=10 + rank, with_prior=True)
        local = _local_grads(cfg, sd, batch, dev)
        torch.manual_seed(1000 + rank)                                 # different init per rank: the broadcast must equalise
        m = flowtron.Flowtron(**cfg).to(dev).eval()
        if rank == 0:
            m.load_state_dict(sd)
        opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)       # optimizer first, wrapper second (train.py:230,251)
        m = D.apply_gradient_allreduce(m)
        crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        res = {"local": local}
        w_before = m._grad_arena.flat_param.detach().cpu().numpy().copy()
        for it in range(2):
            m.zero_grad()
            out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
            nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
            (nll + gl + 0.01 * ctc).backward()
            torch.cuda.synchronize()
            res["log%d" % it] = list(m._grad_bucket_log)
            res["g%d" % it] = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}
        res["w_before"] = w_before
        opt.clip_grad_norm_(1.0)
        opt.step()
        torch.cuda.synchronize()
        res["w_after"] = m._grad_arena.flat_param.detach().cpu().numpy().copy()
        res["shares_arena"] = opt.arena is m._grad_arena
        res["skipped"] = int(opt.skipped_steps)
        from flowtron_amd import ops as _ops
        res["persist_launches"] = int(_ops.PERSIST_LAUNCHES)
        st = _ops._PERSIST.get(dev)
        res["persist_failures"] = int(st.failures) if st is not None else 0
        res["backend"] = torch.distributed.get_backend()
        res["loss"] = float(D.reduce_tensor(nll.detach(), world))
        q.put((rank, res))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:                                             # the parent must not wait for the queue timeout
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def _run(world, mode, full_width=False, overlap="0", backend="nccl", shared_gpu=False, buckets=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, full_width, overlap, backend, shared_gpu, buckets)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for r in out.values():
        assert "error" not in r, r.get("error")
    return out


def _close(a, b, tol):
    import numpy as np
    return float(np.abs(a - b).max()) <= tol * (float(np.abs(b).max()) + 1e-12) + 1e-9


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_rccl_single_rank_bucketed_allreduce_is_the_identity(mode):
    """FLOWTRON_DP_OVERLAP=1: buckets handed to RCCL from the gradient hooks, under the remaining backward"""
    out = _run(1, mode, overlap="1")[0]
    assert out["backend"] == "nccl" and out["shares_arena"]
    for it in (0, 1):
        log = out["log%d" % it]          # last flow first; the encoder and the embeddings (read by every flow) complete last
        assert log[:2] == ["flows.1", "flows.0"] and sorted(log[2:]) == ["encoder", "speaker_embedding+embedding"], log
        for k, g in out["g%d" % it].items():
            # AVG over one rank; split-K atomics make run-to-run bits differ slightly in the weight gradients
            assert _close(g, out["local"][k], 2e-5 if mode == "f32" else 2e-3), (k, it)
    import numpy as np
    assert np.isfinite(out["w_after"]).all() and np.abs(out["w_after"] - out["w_before"]).max() > 0


@pytest.mark.parametrize("buckets,expect", [(None, ["all"]), ("flow", ["speaker_embedding+embedding", "flows.0", "flows.1", "encoder"])])
def test_rccl_default_regime_is_one_allreduce_at_the_end_of_backward_full_width(buckets, expect):
    """Default regime at H = 1024, bf16 (the step goes through lstm_persist_{fwd,bwd}_k, whole-chip co-resident grids): no
    collective is in flight beside them -- ONE all-reduce of the whole arena leaves from the end-of-backward callback behind the
    poison check of the persistent status word (ft_poison_if_nonzero), and the gradients equal the unwrapped module's.
    FLOWTRON_DP_BUCKETS=flow keeps the round-3 behaviour (the per-flow buckets back to back in arena order) for A/B runs."""
    out = _run(1, "bf16", full_width=True, buckets=buckets)[0]
    for it in (0, 1):
        assert out["log%d" % it] == expect, out["log%d" % it]
        for k, g in out["g%d" % it].items():
            assert _close(g, out["local"][k], 2e-3), (k, it)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_rccl_two_ranks_average_gradients():
    out = _run(2, "f32")
    r0, r1 = out[0], out[1]
    import numpy as np
    assert np.array_equal(r0["w_before"], r1["w_before"])               # start-up broadcast from rank 0
    for k in r0["g0"]:
        assert np.array_equal(r0["g0"][k], r1["g0"][k]), k               # every rank holds the same reduced arena
        assert _close(r0["g0"][k], 0.5 * (r0["local"][k] + r1["local"][k]), 2e-5), k
    assert r0["log0"] == ["all"] and r0["log0"] == r1["log0"]
    assert np.array_equal(r0["w_after"], r1["w_after"]) and abs(r0["loss"] - r1["loss"]) < 1e-6


@pytest.mark.skipif(os.environ.get("FLOWTRON_TEST_SHARED_GPU", "1") != "1", reason="FLOWTRON_TEST_SHARED_GPU=0")
def test_two_ranks_on_one_gpu_over_gloo_with_the_persistent_kernels():
    """RCCL refuses two ranks on one device, so the N > 1 code path on real HIP streams runs over gloo: two processes on cuda:0,
    full width (H 1024, bf16: every step launches the whole-chip persistent recurrences), default regime (one all-reduce of the
    arena at the end of backward, behind ft_poison_if_nonzero), different utterances per rank.  Two processes' persistent grids
    may meet on the chip; then their bounded spins time out, the status word poisons that step on BOTH ranks (the NaN
    travels through the all-reduce) and the fused RAdam drops it.  Either way: no hang, every rank holds the same reduced
    arena and the same weights afterwards; when no launch failed, the arena is the mean of the two local gradients."""
    out = _run(2, "bf16", full_width=True, backend="gloo", shared_gpu=True)
    r0, r1 = out[0], out[1]
    import numpy as np
    assert r0["backend"] == "gloo" and np.array_equal(r0["w_before"], r1["w_before"])
    assert r0["log0"] == r1["log0"] == ["all"]
    clean = True
    for k in r0["g1"]:
        a, b = r0["g1"][k], r1["g1"][k]
        assert np.array_equal(a, b, equal_nan=True), k                  # every rank holds the same reduced arena
        clean &= bool(np.isfinite(a).all())
    if clean:
        for k in r0["g1"]:
            assert _close(r0["g1"][k], 0.5 * (r0["local"][k] + r1["local"][k]), 4e-3), k
        assert r0["skipped"] == r1["skipped"] == 0
    else:
        assert r0["skipped"] == r1["skipped"] == 1                       # the poisoned step was dropped on both ranks
    assert np.array_equal(r0["w_after"], r1["w_after"]) and np.isfinite(r0["w_after"]).all()
    print("\n[two ranks, one GPU, gloo] clean=%s skipped=%s persistent launches %s / %s" %
          (clean, r0["skipped"], r0["persist_launches"], r1["persist_launches"])
          + "  failures %s / %s" % (r0["persist_failures"], r1["persist_failures"]))


def _busy_worker(port, q, overlap):
    """world-size-1 RCCL job at full width (bf16: every step launches the whole-chip persistent recurrences) with a FOREIGN kernel
    holding eight CUs for 0.8 s across one step -- what an in-flight collective of another stream does to a persistent grid."""
    try:
        import warnings
        for p_ in (ROOT, HERE):
            if p_ not in sys.path:
                sys.path.insert(0, p_)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLOWTRON_MFMA="bf16", LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0", FLOWTRON_DP_OVERLAP=overlap, FLOWTRON_LSTM_PERSIST="1")
        import distributed as D
        import flowtron
        from flowtron_amd import _lib as L
        from flowtron_amd import ops
        from flowtron_amd.optim import RAdam
        from oracle import synth
        D.init_distributed(0, 1, "nccl", None)
        dev = torch.device("cuda", 0)
        cfg = dict(synth.DEFAULT_MODEL_CONFIG)
        m = flowtron.Flowtron(**cfg).to(dev).eval()
        m.load_state_dict(synth.make_state_dict(cfg, seed=3))
        opt = RAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
        m = D.apply_gradient_allreduce(m)
        crit = flowtron.FlowtronLoss(1.0, False, True, True, 0.01, -8)
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.make_batch(cfg, [40, 33, 21], [12, 9, 7], seed=10, with_prior=True).items()}
        side = torch.cuda.Stream()
        res = {"usable_before": bool(ops.persist_usable(dev)), "losses": [], "params_finite": []}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for it in range(6):
                if it == 2:                                  # eight CUs taken for 0.8 s (the persistent kernels give up after 0.5 s)
                    torch.cuda.synchronize()
                    L.check(L.lib().ft_debug_hold_cus(8, 80000000, side.cuda_stream), "hold")
                m.zero_grad()
                out = m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"], b["attn_prior"])
                nll, gl, ctc = crit(out, b["gate_target"], b["in_lens"], b["out_lens"])
                (nll + gl + 0.01 * ctc).backward()
                opt.clip_grad_norm_(1.0)
                opt.step()
                torch.cuda.synchronize()
                res["losses"].append(float(nll))
                res["params_finite"].append(bool(torch.isfinite(m._grad_arena.flat_param).all()))
                if it == 1:
                    res["launches_clean"] = int(ops.PERSIST_LAUNCHES)
        res["warnings"] = [str(w.message) for w in caught if "persistent recurrence" in str(w.message)]
        res["skipped"] = int(opt.skipped_steps)
        res["usable_after"] = bool(ops.persist_usable(dev))
        res["failures"] = int(ops._PERSIST[dev].failures)
        res["status"] = int(ops.persist_status(dev).item())
        res["launches_end"] = int(ops.PERSIST_LAUNCHES)
        q.put((0, res))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((0, {"error": traceback.format_exc()}))


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_persistent_recurrences_beside_a_foreign_kernel_time_out_once_and_the_run_continues(overlap):
    """Multi-GPU readiness without a multi-GPU box: in an N > 1 job a collective kernel of another stream may hold
    CUs while a step's whole-chip persistent recurrences are launched.  Here a stand-in (ft_debug_hold_cus: eight workgroups that
    each take a whole CU for 0.8 s, on a second stream) does exactly that across one step of a world-size-1 RCCL job, in both DP
    regimes: the persistent grid is not co-resident, its bounded spins give up (0.5 s), the status word poisons that step's
    gradients in front of the all-reduce, the fused RAdam drops the update, the host warns ONCE and switches the device to the
    launch-per-step kernels, and the loss sequence continues on them -- finite weights throughout, no hang, no exception."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_busy_worker, args=(_free_port(), q, overlap))
    pr.start()
    _, r = q.get(timeout=500)
    pr.join(timeout=120)
    assert "error" not in r, r.get("error")
    if not r["usable_before"]:
        pytest.skip("persistent recurrences not usable on this device")
    assert len(r["warnings"]) == 1, r["warnings"]
    assert r["failures"] == 1 and not r["usable_after"] and r["status"] == 0
    assert 1 <= r["skipped"] <= 2, r["skipped"]                  # the step beside the foreign kernel (+ at most the one after it)
    assert all(r["params_finite"]) and all(math.isfinite(x) for x in r["losses"]), r
    assert r["launches_clean"] > 0 and r["losses"][-1] != r["losses"][2]          # training went on after the dropped step
    print("\n[foreign kernel, overlap=%s] losses %s skipped %d persistent launches %d -> %d" %
          (overlap, ["%.4f" % x for x in r["losses"]], r["skipped"], r["launches_clean"], r["launches_end"]))


def test_bench_script_with_two_ranks_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2` -- the driver's own command for the
    scaling runs -- on this one-GPU box through bench.py's test hook (both ranks on cuda:0, gloo instead of RCCL): the barriers,
    the MAX-over-ranks time, the SUM of frames, one JSON line from rank 0, a clean exit of both ranks."""
    import json
    import subprocess
    env = dict(os.environ, BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-infer", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 64 and d["config"]["valid_frames_per_step"] > 18932        # both ranks' frames
    assert abs(d["value"] - d["config"]["valid_frames_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert "ONE in-place RCCL all-reduce" in d["config"]["workload"] and "at the end of backward" in d["config"]["workload"]
    assert "dominant_kernel" in d["roofline"]
    # the self-diagnosing N > 1 block: ranks, backend, one collective per step, its time on every rank
    dp = d["dp"]
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["collectives_per_step"] == 1 and dp["regime"] == "end-of-backward"
    assert len(dp["allreduce_ms_per_rank"]) == 2 and all(v > 0 for v in dp["allreduce_ms_per_rank"])
    assert dp["exposed_comm_ms"] == max(dp["allreduce_ms_per_rank"]) and dp["arena_mb"] > 200
