"""The reference's training / inference call sequence (tests/ref_loop.py = train.py:205-377, inference.py:40-71 restated call
for call) on the drop-in modules, on the MI355X: fp32, bf16 operands, and `fp16_run` (GradScaler + torch clip_grad_norm_ on the
arena views), through a real DataLoader worker with device-side mel / prior, a train.py-format checkpoint (pickled module +
optimizer state) at iteration 2, a resume from it, and inference from the checkpoint."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (ROOT, HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

SMALL = dict(n_text=64, n_text_dim=128, n_speaker_dim=32, n_attn_channels=64, n_hidden=128)


def _run(tmp_path, monkeypatch, mode, fp16_run, model_overrides, iters=4, checkpoint_path=""):
    import ref_fixture
    import ref_loop
    monkeypatch.setenv("FLOWTRON_MFMA", mode)
    monkeypatch.setenv("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")          # train.py:112 torch.load of a pickled module
    root = tmp_path / ("data_%s_%d" % (mode, int(fp16_run)))
    cfg_path, cfg = ref_fixture.make_config(str(root), model_overrides=model_overrides,
                                            train_overrides=dict(fp16_run=fp16_run, checkpoint_path=checkpoint_path))
    return ref_loop.train(cfg, iters), cfg


# ("auto", True): FLOWTRON_MFMA unset/auto + the loop's amp.autocast(float16) -> the fp16 operand kernels, as in the reference's AMP run
@pytest.mark.parametrize("mode,fp16_run", [("f32", False), ("bf16", False), ("bf16", True), ("f16", True), ("auto", True)])
def test_training_loop_checkpoint_resume_and_inference(tmp_path, monkeypatch, mode, fp16_run):
    import ref_loop
    out, cfg = _run(tmp_path, monkeypatch, mode, fp16_run, SMALL, iters=5)
    losses = out["losses"]
    assert sorted(losses) == [0, 1, 2, 3, 4] and all(torch.isfinite(torch.tensor(v)) for v in losses.values())
    assert sorted(out["val_losses"]) == [0, 2, 4] and len(out["checkpoints"]) == 3
    if fp16_run:
        assert out["scale"] > 1.0                       # GradScaler was live (65536 unless an inf step halved it)
    # the optimizer really stepped: RAdam state carries the iteration count, moments are non-zero views of the arenas
    opt = out["optimizer"]
    assert opt._step == 5 and float(opt.flat_m.abs().sum()) > 0
    # resume from the iteration-2 checkpoint (train.py:242-246): same data order (seeded loader) -> iterations 3, 4 reproduce
    ck = out["checkpoints"][1]
    cfg2 = copy.deepcopy(cfg)
    cfg2["train_config"]["checkpoint_path"] = ck
    torch.manual_seed(0)
    res = ref_loop.train(cfg2, 5)
    assert sorted(res["losses"]) == [3, 4]
    assert res["optimizer"]._step == 5                   # 3 restored + 2 new: the moments came back from the checkpoint
    # inference.py path from the checkpoint file
    monkeypatch.setenv("FLOWTRON_MFMA", "f32")
    mels, attentions = ref_loop.infer(cfg, out["checkpoints"][-1], "the quick brown fox", 0, 40, 0.5, 1.0, 1234)
    assert mels.shape == (1, 80, 40) and torch.isfinite(mels).all()
    assert len(attentions) == cfg["model_config"]["n_flows"] and len(attentions[0]) == 40


def test_full_width_loop_runs_the_persistent_recurrences(tmp_path, monkeypatch):
    """default model_config (H = 1024) in bf16 mode, 3 iterations of the loop: the step goes through the persistent LSTM kernels
    (B = 4 <= 32) and the status word stays clean; losses finite and decreasing is not required (3 steps), only sanity."""
    from flowtron_amd import ops
    out, _ = _run(tmp_path, monkeypatch, "bf16", True, {}, iters=3)
    ops.check_persist_status()
    assert all(torch.isfinite(torch.tensor(v)) for v in out["losses"].values())


def test_five_iteration_trajectory_matches_the_real_reference_loop(monkeypatch):
    """SURVEY 8a row a25 / VERDICT r3 missing #2: the composed loop of train.py:282-331 -- model.zero_grad(), forward, FlowtronLoss,
    backward, torch.nn.utils.clip_grad_norm_, RAdam.step, five iterations on two alternating ragged batches -- on the drop-in
    modules (fp32 MFMA mode) against `tests/golden/train_traj.pt`, which tests/golden/make_golden_r4.py produced by running the
    SAME function (`run_reference_loop`) over the REAL reference's Flowtron / FlowtronLoss / radam.RAdam on CPU with dropout
    neutralised: the four losses and the pre-clip gradient norm of every iteration, every parameter after the last one."""
    import importlib.util
    import flowtron
    import radam
    monkeypatch.setenv("FLOWTRON_MFMA", "f32")
    spec_ = importlib.util.spec_from_file_location("_make_golden_r4", os.path.join(HERE, "golden", "make_golden_r4.py"))
    mg = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mg)
    g = torch.load(os.path.join(HERE, "golden", "train_traj.pt"), weights_only=False)
    assert g["spec"] == mg.TRAJ, "the committed fixture was made from another specification"

    def ones_masks(model, b):                    # the drop-in encoder draws keep-masks itself (no F.dropout to patch): all kept, unscaled
        Lk, B = b["text"].shape[1], b["text"].shape[0]
        C = model.embedding.weight.shape[1]
        model.encoder.dropout_masks = [torch.ones(Lk, B, C, device="cuda") for _ in range(3)]

    res = mg.run_reference_loop(flowtron.Flowtron, flowtron.FlowtronLoss, radam.RAdam, device="cuda", neutralise_dropout=False,
                                prepare=ones_masks)
    torch.cuda.synchronize()
    for it in range(g["spec"]["iters"]):
        for j, name in enumerate(("loss", "gate", "nll", "ctc")):
            r, m = float(g["losses"][it, j]), float(res["losses"][it, j])
            assert abs(m - r) < 1e-4 * max(1.0, abs(r)), (it, name, m, r)
        rn, mn = float(g["grad_norms"][it]), float(res["grad_norms"][it])
        assert abs(mn - rn) < 5e-4 * rn, (it, mn, rn)
    from oracle import synth
    init = synth.make_state_dict(g["spec"]["cfg"], seed=g["spec"]["seed"])
    worst = ("", 0.0)
    for k, ref in g["params"].items():
        moved = (ref - init[k]).abs().max().item()
        err = (res["params"][k] - ref).abs().max().item()
        if err > worst[1]:
            worst = (k, err)
        assert err < 2e-6 + 1e-2 * moved, (k, err, moved)
    assert res["optimizer"]._step == g["spec"]["iters"] and res["optimizer"].skipped_steps == 0
    print("trajectory: worst parameter deviation %.2e (%s)" % (worst[1], worst[0]))
