"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every symbol include/flowtron_hip.h declares,
the host mirror keeps the reference's interface (state_dict layout, config schema, pickling), the product path
refuses CPU tensors (no fallback), and the pure-host pieces (batched attention-CTC, RAdam step size) match the
oracle / the reference formula."""
import ctypes
import json
import os
import pickle
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from flowtron_amd import _lib
    from flowtron_amd import build
    lib = build.build(verbose=False)
    h = ctypes.CDLL(lib)
    hdr = open(os.path.join(ROOT, "include", "flowtron_hip.h")).read()
    declared = set(re.findall(r"\b(ft_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    for name in declared:
        assert hasattr(h, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert h.ft_abi_version() == 14


def test_state_dict_layout_matches_reference_spec():
    import flowtron
    from oracle import synth
    for cfg in (synth.DEFAULT_MODEL_CONFIG, dict(synth.DEFAULT_MODEL_CONFIG, n_flows=3, n_speakers=123), synth.SMALL_MODEL_CONFIG,
                dict(synth.SMALL_MODEL_CONFIG, use_cumm_attention=True)):
        m = flowtron.Flowtron(**cfg)
        sd = m.state_dict()
        spec = synth.state_dict_spec(cfg)
        assert list(sd.keys()) == [k for k, _ in spec]
        for k, shp in spec:
            assert tuple(sd[k].shape) == tuple(shp), k
    m = flowtron.Flowtron(**synth.DEFAULT_MODEL_CONFIG)
    assert sum(p.numel() for p in m.parameters()) == 60977473          # 243.9 MB fp32
    for k, p in m.named_parameters():                                   # flowtron.py:651-653 zero init
        if k.endswith("conv.weight") and "convolutions" not in k:
            assert float(p.abs().max()) == 0.0


def test_reference_config_json_schema_is_accepted():
    """config.json's model_config keys are splatted into the ctor (train.py:221): the key set IS the signature."""
    ref_cfg = os.path.join("/root/reference", "config.json")
    if not os.path.exists(ref_cfg):
        pytest.skip("reference not mounted")
    import flowtron
    cfg = json.load(open(ref_cfg))
    m = flowtron.Flowtron(**cfg["model_config"])
    assert len(m.flows) == cfg["model_config"]["n_flows"]
    t = cfg["train_config"]
    flowtron.FlowtronLoss(t["sigma"], bool(cfg["model_config"]["n_components"]), t["gate_loss"], t["use_ctc_loss"],
                          t["ctc_loss_weight"], t["blank_logprob"])


def test_checkpoint_pickle_roundtrip(tmp_path, monkeypatch):
    """train.py:131-139 pickles the whole module as flowtron.Flowtron; inference.py:54 reads 'state_dict'.  torch >= 2.6
    needs TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 for train.py:112's plain torch.load; importing the
    package must NOT set it process-wide."""
    import flowtron
    assert os.environ.get("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD") is None, "the package must not change torch.load's default"
    monkeypatch.setenv("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")
    from oracle import synth
    m = flowtron.Flowtron(**synth.SMALL_MODEL_CONFIG)
    p = tmp_path / "model_0"
    torch.save({"model": m, "iteration": 7, "learning_rate": 1e-3}, p)
    ck = torch.load(p, map_location="cpu")
    assert type(ck["model"]).__module__ == "flowtron" and type(ck["model"]).__name__ == "Flowtron"
    m2 = flowtron.Flowtron(**synth.SMALL_MODEL_CONFIG)
    m2.load_state_dict(ck["model"].state_dict())
    assert pickle.loads(pickle.dumps(m)).state_dict().keys() == m.state_dict().keys()


def test_cpu_tensors_are_refused_not_emulated():
    import flowtron
    from oracle import synth
    cfg = synth.SMALL_MODEL_CONFIG
    m = flowtron.Flowtron(**cfg)
    b = synth.make_batch(cfg, [9, 5], [4, 3], seed=1, with_prior=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(b["mel"], b["speaker_ids"], b["text"], b["in_lens"], b["out_lens"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.infer(torch.zeros(1, 80, 4), b["speaker_ids"][:1], b["text"][:1])


def test_product_package_never_imports_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "flowtron_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    for f in ("flowtron.py", "audio_processing.py", "distributed.py", "radam.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def _batched_ctc_restatement(attn_logprob, in_lens, out_lens, blank_logprob):
    """The batching the HIP kernel pair (ft_attn_ctc_fwd/bwd) implements, in torch ops on CPU tensors: blank prepended,
    classes beyond in_len masked, one log-softmax, one batched ctc_loss with targets 1..K, per-sample mean over K."""
    B, T, Lk = attn_logprob.shape
    x = torch.nn.functional.pad(attn_logprob, (1, 0), value=blank_logprob)             # [B,T,L+1], blank first
    cls = torch.arange(Lk + 1)[None, None, :]
    x = x.masked_fill(cls > in_lens[:, None, None], -1.0e4)   # exp() underflows to exactly 0: same softmax, finite grads
    lp = torch.log_softmax(x, dim=2).transpose(0, 1)                                   # [T,B,L+1]
    targets = torch.arange(1, Lk + 1)[None, :].expand(B, -1)
    loss = torch.nn.functional.ctc_loss(lp, targets, input_lengths=out_lens, target_lengths=in_lens, blank=0,
                                        reduction="none", zero_infinity=True)
    return (loss / in_lens.to(loss.dtype)).mean()


def test_batched_attention_ctc_matches_per_sample_loop():
    from oracle import flowtron_oracle as O
    torch.manual_seed(0)
    B, T, Lk = 4, 23, 9
    in_lens = torch.tensor([9, 7, 7, 3])
    out_lens = torch.tensor([23, 20, 11, 9])
    lp = torch.log_softmax(torch.randn(B, T, Lk), 2).requires_grad_(True)
    ref = O.attention_ctc_loss(lp, in_lens, out_lens, blank_logprob=-8)
    ref.backward()
    g_ref = lp.grad.clone()
    lp.grad = None
    mine = _batched_ctc_restatement(lp, in_lens, out_lens, -8)
    mine.backward()
    assert abs(mine.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert (lp.grad - g_ref).abs().max().item() < 1e-6
    assert torch.isfinite(lp.grad).all()


def test_radam_step_size_matches_reference_formula():
    from flowtron_amd.optim import RAdam
    ref_path = "/root/reference/radam.py"
    for step in (1, 2, 5, 6, 7, 100, 10000):
        ss, rect = RAdam.step_size_for(step, 1e-3, 0.9, 0.999)
        beta2_t = 0.999 ** step
        n_max = 2 / (1 - 0.999) - 1
        n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
        assert rect == (n_sma >= 5)
        if not rect:
            assert abs(ss - 1e-3 / (1 - 0.9 ** step)) < 1e-12
    if os.path.exists(ref_path):       # run the real reference optimizer for a few steps on one tensor
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_radam", ref_path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        import warnings
        p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 3.0]))
        opt = mod.RAdam([p], lr=1e-3, weight_decay=1e-6)
        q, m, v = p.detach().clone().double(), torch.zeros(3).double(), torch.zeros(3).double()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for step in range(1, 9):
                g = torch.tensor([0.3, -0.1, 0.2]) * step
                p.grad = g.clone()
                opt.step()
                ss, rect = RAdam.step_size_for(step, 1e-3, 0.9, 0.999)
                v = 0.999 * v + 0.001 * g.double() ** 2
                m = 0.9 * m + 0.1 * g.double()
                q = q - 1e-6 * 1e-3 * q
                q = q - ss * m / (v.sqrt() + 1e-8) if rect else q - ss * m
                assert (p.detach().double() - q).abs().max().item() < 1e-6, step


def test_mel_filterbank_and_window_match_oracle():
    from flowtron_amd.audio import hann_window, slaney_mel_filterbank
    from oracle import flowtron_oracle as O
    fb = torch.from_numpy(slaney_mel_filterbank(22050, 1024, 80, 0.0, 8000.0))
    assert (fb - O.mel_filterbank()).abs().max().item() < 1e-7
    assert fb.shape == (80, 513) and float(fb.min()) >= 0
    assert (torch.from_numpy(hann_window(1024, 1024)) - O.hann_periodic(1024).float()).abs().max().item() < 1e-7


def test_mel_filterbank_pinned_to_published_librosa_equivalent(golden_dir):
    """audio_processing.py:104-107 takes the 80 x 513 filterbank from librosa, which is not installed here.  The fixture holds
    the output of transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney') -- Hugging Face's published
    replacement for librosa.filters.mel (tests/golden/make_golden_fb.py) -- and pins BOTH restatements (oracle and product) to
    it; when transformers is importable the fixture itself is regenerated and compared, so it cannot drift silently."""
    import numpy as np
    from flowtron_amd.audio import slaney_mel_filterbank
    from oracle import flowtron_oracle as O
    g = np.load(os.path.join(golden_dir, "mel_fb_hf_slaney.npz"))
    ref = g["fb"]
    assert ref.shape == (80, 513) and ref.dtype == np.float64
    prod = np.asarray(slaney_mel_filterbank(22050, 1024, 80, 0.0, 8000.0), dtype=np.float64)
    orac = O.mel_filterbank().double().numpy()
    assert np.abs(prod - ref).max() < 5e-9 and np.abs(orac - ref).max() < 5e-9      # values up to 0.0265: < 2e-7 relative
    # structure librosa guarantees: every filter non-empty, Slaney area normalisation, at most two filters per bin
    assert (ref.sum(1) > 0).all() and ((ref > 0).sum(0) <= 2).all()
    try:
        from transformers.audio_utils import mel_filter_bank
    except Exception:
        return
    again = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                            sampling_rate=22050, norm="slaney", mel_scale="slaney").T
    assert np.abs(again - ref).max() == 0.0


def _reference_class(path, name):
    """Extract ONE class from a reference module without importing the module (data.py pulls in librosa etc.)."""
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def test_datacollate_matches_reference_wire_format():
    from flowtron_amd.data import DataCollate
    ref_path = "/root/reference/data.py"
    torch.manual_seed(3)
    batch = []
    for t, l in ((37, 9), (52, 14), (20, 14), (45, 3)):
        batch.append((torch.randn(80, t), torch.tensor([t % 3]), torch.randint(0, 100, (l,)), torch.rand(t, l)))
    mine = DataCollate(1, True)(batch)
    assert mine[0].shape == (4, 80, 52) and mine[2].shape == (4, 14) and mine[6].shape == (4, 52, 14)
    assert mine[3].tolist() == sorted(mine[3].tolist(), reverse=True)
    assert all(float(mine[5][i, int(mine[4][i]) - 1]) == 1.0 and float(mine[5][i, : int(mine[4][i]) - 1].sum()) == 0.0 for i in range(4))
    if os.path.exists(ref_path):
        ref = _reference_class(ref_path, "DataCollate")(1, True)(batch)
        for a, b in zip(mine, ref):
            assert torch.equal(a, b)
        ref2 = _reference_class(ref_path, "DataCollate")(4, False)([x[:3] for x in batch])
        mine2 = DataCollate(4, False)([x[:3] for x in batch])
        assert mine2[6] is None and ref2[6] is None
        for a, b in zip(mine2[:6], ref2[:6]):
            assert torch.equal(a, b)


def test_length_bucket_batch_sampler_partitions_and_cuts_padding():
    """batches of similar mel length.  Every utterance is used at most once per epoch, all ranks get the
    same number of batches, epochs differ, and the padded-frame fraction drops well below random batching."""
    from flowtron_amd.data import LengthBucketBatchSampler
    g = torch.Generator().manual_seed(0)
    lengths = (torch.randn(5000, generator=g) * 190 + 566).clamp(100, 862).round().long()
    B, W = 32, 4
    seen, per_rank = [], []
    for r in range(W):
        s = LengthBucketBatchSampler(lengths, B, rank=r, world_size=W, seed=7)
        batches = list(s)
        assert len(batches) == len(s) == (5000 // B) // W and all(len(b) == B for b in batches)
        per_rank.append(batches)
        seen += [i for b in batches for i in b]
    assert len(seen) == len(set(seen))                                   # disjoint across ranks and batches
    valid = sum(int(lengths[b].sum()) for bs in per_rank for b in bs)
    padded = sum(int(lengths[b].max()) * B for bs in per_rank for b in bs)
    rnd = torch.randperm(5000, generator=g)[:4992].view(-1, B)
    frac_rnd = float(lengths[rnd].sum()) / float((lengths[rnd].max(1).values * B).sum())
    assert valid / padded > 0.93 and frac_rnd < 0.75, (valid / padded, frac_rnd)
    # same step, all ranks: similar cost -- the slowest rank's T_max summed over the epoch is within 8 % of the mean rank's
    slow = mean = 0.0
    for k in range(len(per_rank[0])):
        mx = [int(lengths[per_rank[r][k]].max()) for r in range(W)]
        slow += max(mx)
        mean += sum(mx) / W
    assert slow <= 1.08 * mean, (slow, mean)
    s0 = LengthBucketBatchSampler(lengths, B, seed=7)
    e0 = list(s0)
    s0.set_epoch(1)
    assert list(s0) != e0 and list(LengthBucketBatchSampler(lengths, B, seed=7)) == e0


def test_workspace_size_queries_run_without_a_gpu():
    """Pure host functions of the C ABI (no kernel launch, no device memory): the geometry they promise to callers."""
    import ctypes as C
    from flowtron_amd import _lib as L
    lib = L.lib()
    up = lambda v, m: (v + m - 1) // m * m
    for rows, cols in ((1, 1), (27584, 4096), (300, 80), (4096, 1664)):
        assert lib.ft_bf16_image_bytes(rows, cols) == up(up(rows + 32, 256) * up(cols, 256) * 2, 256)
    assert lib.ft_bf16_image_bytes(0, 5) == 0
    # ft_gemm_workspace_bytes: zero for fp32 mode / batched / small problems, image bytes of both operands otherwise
    a = L.GemmArgs(None, None, None, None, 27584, 4096, 1664, 1, 1664, 1, 1, 1664, 4096, 0, 0, 0, 1.0, 0.0, 0, L.FT_BF16, 0, None, 0)
    need = lib.ft_gemm_workspace_bytes(C.byref(a))
    assert need == up(up(27584, 256) * up(1664, 32) * 2, 256) + up(up(4096, 256) * up(1664, 32) * 2, 256)
    a.mode = L.FT_F32
    assert lib.ft_gemm_workspace_bytes(C.byref(a)) == 0
    a.mode, a.batch = L.FT_BF16, 4
    assert lib.ft_gemm_workspace_bytes(C.byref(a)) == 0
    a.batch, a.M, a.N, a.K = 1, 16, 16, 16
    assert lib.ft_gemm_workspace_bytes(C.byref(a)) == 0
    assert lib.ft_lstm2_supported(32, 1024) == 1 and lib.ft_lstm2_supported(65, 1024) == 0 and lib.ft_lstm2_supported(8, 100) == 0
    assert lib.ft_lstm_bidir_supported(32, 256) == 1 and lib.ft_lstm_bidir_supported(32, 96) == 0
    assert lib.ft_lstm2_workspace_bytes(32, 1024) > 3 * 4 * 1024 * 1024 * 2          # three bf16 weight images + state
    assert lib.ft_attn_ctc_workspace_floats(2, 10, 5) == 2 * 2 * 10 * 11 + 2 * 10 + 2


def test_radam_load_state_dict_restores_the_flat_arenas():
    """train.py:123 resume: after load_state_dict the fused kernel's arenas (flat_m / flat_v / _step) hold the checkpoint's
    moments and state[p] are views into them again.  Host logic only -- no kernel."""
    from flowtron_amd.optim import RAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(1)), torch.nn.Parameter(torch.randn(33))]
    o = RAdam(ps, lr=1e-3)
    o.flat_m.normal_()
    o.flat_v.uniform_()
    o._step = 4
    for p in ps:
        o.state[p]["step"] = 4
    sd = o.state_dict()
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o2 = RAdam(qs, lr=1e-3)
    o2.load_state_dict(sd)
    assert o2._step == 4
    for p, q in zip(ps, qs):
        assert torch.equal(o2.state[q]["exp_avg"], o.state[p]["exp_avg"])
        assert torch.equal(o2.state[q]["exp_avg_sq"], o.state[p]["exp_avg_sq"])
        lo = o2.flat_m.data_ptr()
        assert lo <= o2.state[q]["exp_avg"].data_ptr() < lo + o2.flat_m.numel() * 4
        assert o2.state[q]["step"] == 4
    off = o2.arena.offsets
    assert torch.equal(o2.flat_m[off[2]:off[2] + 33], o.flat_m[o.arena.offsets[2]:o.arena.offsets[2] + 33])


def test_weight_gradients_land_in_the_arena_without_a_copy():
    """dist.FlatArena.grad_view_for_pass / arena_slot (host logic; CPU tensors): in a backward pass that starts with every .grad None
    the arena is zeroed once and a weight-gradient producer gets its parameter's arena slice -- autograd adopts the returned view as
    .grad (no copy: the address lies inside the arena) and adopt_stray_grads finds nothing to move; a reshape of a parameter is
    recognised, a slice of it is not; a second contribution to the same parameter and a pass that starts with gradients in place
    (accumulation) get None and accumulate through autograd as before."""
    from flowtron_amd import dist as D
    from flowtron_amd import ops
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(6, 4)), torch.nn.Parameter(torch.randn(3, 2, 2)), torch.nn.Parameter(torch.randn(5))]
    arena = D.FlatArena(ps)
    lo, hi = arena._ptr_lo, arena._ptr_hi
    assert D.arena_slot(ps[0]) == (arena, 0) and D.arena_slot(ps[1].reshape(3, 4)) == (arena, 1)
    assert D.arena_slot(ps[0][:3]) is None and D.arena_slot(torch.randn(6, 4)) is None

    class MulW(torch.autograd.Function):                      # y = x * W, dW through a zero-filled accumulating buffer like the GEMMs'
        @staticmethod
        def forward(ctx, x, W):
            ctx.save_for_backward(x, W)
            return x * W

        @staticmethod
        def backward(ctx, g):
            x, W = ctx.saved_tensors
            slot = D.arena_slot(W)
            dW = slot[0].grad_view_for_pass(slot[1], ops._current_graph_task(), W.shape) if slot else None
            served.append(dW is not None)
            if dW is None:
                dW = torch.zeros_like(W)
            dW += g * x
            return None, dW

    x0, x1, x2 = torch.randn(6, 4), torch.randn(3, 4), torch.randn(5)
    served = []
    arena.flat_grad.fill_(7.0)                                # stale values of an earlier step
    arena.zero_grad()
    (MulW.apply(x0, ps[0]).sum() + MulW.apply(x1, ps[1].reshape(3, 4)).sum()).backward()
    assert served == [True, True] or served == [True, True][::-1]
    assert all(lo <= p.grad.data_ptr() < hi for p in ps[:2]) and ps[2].grad is None       # adopted as they are: no copy
    assert torch.allclose(ps[0].grad, x0) and torch.allclose(ps[1].grad.reshape(3, 4), x1)
    skipped = arena.adopt_stray_grads(copy=True)
    assert not arena.adopt_copied and skipped == [(arena.offsets[2], 5)]
    off = arena.offsets
    assert torch.allclose(arena.flat_grad[off[0]:off[0] + 24].view(6, 4), x0) and float(arena.flat_grad[off[2]:off[2] + 5].abs().max()) == 0.0
    # a second pass WITHOUT zero_grad accumulates: nothing is served, the arena is not zeroed again
    served = []
    MulW.apply(x0, ps[0]).sum().backward()
    assert served == [False] and torch.allclose(ps[0].grad, 2 * x0) and lo <= ps[0].grad.data_ptr() < hi
    # a parameter read twice in one pass: the second request is not served; whatever tensor autograd forms of the two contributions,
    # the adoption leaves their SUM in the arena (the engine adds contributions of shared storage out of place: that one is copied)
    served = []
    arena.zero_grad()
    (MulW.apply(x2, ps[2]).sum() + MulW.apply(2 * x2, ps[2]).sum() + MulW.apply(x0, ps[0]).sum()).backward()
    assert sorted(served) == [False, True, True]
    arena.adopt_stray_grads(copy=True)
    assert torch.allclose(arena.flat_grad[off[2]:off[2] + 5], 3 * x2) and torch.allclose(arena.flat_grad[off[0]:off[0] + 24].view(6, 4), x0)
    assert all(lo <= p.grad.data_ptr() < hi for p in (ps[0], ps[2]))


def test_gradient_buckets_cover_the_arena_flow_by_flow():
    """dist.gradient_buckets on the default model: contiguous, exhaustive, one bucket per flow (110.7 MB each) between the
    embeddings and the encoder -- the ranges the data-parallel wrapper hands to RCCL one by one."""
    import flowtron
    from flowtron_amd import dist as D
    from oracle import synth
    m = flowtron.Flowtron(**synth.DEFAULT_MODEL_CONFIG)
    arena = D.FlatArena(list(m.parameters()))
    b = D.gradient_buckets(m, arena)
    assert [n for n, _, _, _ in b] == ["speaker_embedding+embedding", "flows.0", "flows.1", "encoder"]
    assert b[0][1] == 0 and b[-1][2] == arena.numel and all(b[i][2] == b[i + 1][1] for i in range(3))
    assert sum(len(idx) for _, _, _, idx in b) == len(arena.params) == 68
    assert abs((b[1][2] - b[1][1]) * 4 / 1e6 - 110.7) < 0.1


def test_operand_mode_follows_env_then_autocast(monkeypatch):
    from flowtron_amd import _lib as L
    for v, want in (("f32", L.FT_F32), ("bf16", L.FT_BF16), ("f16", L.FT_F16), ("fp16", L.FT_F16)):
        monkeypatch.setenv("FLOWTRON_MFMA", v)
        assert L.mfma_mode() == want
    monkeypatch.setenv("FLOWTRON_MFMA", "int8")
    with pytest.raises(ValueError):
        L.mfma_mode()
    monkeypatch.delenv("FLOWTRON_MFMA")
    assert L.mfma_mode() == L.FT_F32                      # no autocast region: parity mode
    assert L.is16(L.FT_BF16) and L.is16(L.FT_F16) and not L.is16(L.FT_F32)


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: the package and the root drop-ins must not import, load or execute anything from it
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity legs may)."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ("flowtron.py", "audio_processing.py", "distributed.py", "radam.py", "data.py")]
    pkg = os.path.join(root, "flowtron_amd")
    files += [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")]
    for path in files:
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), (path, names)
    for f in os.listdir(os.path.join(pkg, "csrc")):
        assert "oracle" not in open(os.path.join(pkg, "csrc", f), errors="ignore").read().lower() or f.endswith(".md"), f
