"""Call-for-call restatement of the reference's training / inference scripts (train.py:100-139, 140-202, 205-377;
inference.py:40-71) used to drive the drop-in modules on the GPU box, where /root/reference does not exist (the REAL train.py
is exercised as far as possible without a GPU by tests/test_reference_scripts_cpu.py).  Every statement that touches
`flowtron`, `data`, `radam`, `distributed` or torch.cuda.amp is kept in the reference's order; logging / printing is dropped.
TEST INFRASTRUCTURE."""
import os

import torch
from torch.cuda import amp
from torch.utils.data import DataLoader

from data import Data, DataCollate
from flowtron import Flowtron, FlowtronLoss
from radam import RAdam


def grapheme_frontend(text):
    """stand-in text front end for boxes without the reference's `text` package: a-z, space and ' -> ids 1..28."""
    table = {c: i + 1 for i, c in enumerate("abcdefghijklmnopqrstuvwxyz '")}
    return [table[c] for c in text.lower() if c in table]


def prepare_dataloaders(data_config, n_gpus, batch_size, text_frontend):           # train.py:56-81
    ignore_keys = ["training_files", "validation_files"]
    kw = dict((k, v) for k, v in data_config.items() if k not in ignore_keys)
    trainset = Data(data_config["training_files"], text_frontend=text_frontend, **kw)
    valset = Data(data_config["validation_files"], speaker_ids=trainset.speaker_ids, text_frontend=text_frontend, **kw)
    collate_fn = DataCollate(n_frames_per_step=1, use_attn_prior=trainset.use_attn_prior)
    train_loader = DataLoader(trainset, num_workers=1, shuffle=True, sampler=None, batch_size=batch_size, pin_memory=False,
                              drop_last=True, collate_fn=collate_fn)
    return train_loader, valset, collate_fn


def load_checkpoint(checkpoint_path, model, optimizer, ignore_layers=[]):           # train.py:110-128
    assert os.path.isfile(checkpoint_path)
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu")
    iteration = checkpoint_dict["iteration"]
    model_dict = checkpoint_dict["model"].state_dict()
    if len(ignore_layers) > 0:
        model_dict = {k: v for k, v in model_dict.items() if k not in ignore_layers}
        dummy_dict = model.state_dict()
        dummy_dict.update(model_dict)
        model_dict = dummy_dict
    else:
        optimizer.load_state_dict(checkpoint_dict["optimizer"])
    model.load_state_dict(model_dict)
    return model, optimizer, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, filepath, model_config):   # train.py:131-139
    model_for_saving = Flowtron(**model_config).cuda()
    model_for_saving.load_state_dict(model.state_dict())
    torch.save({"model": model_for_saving, "iteration": iteration, "optimizer": optimizer.state_dict(),
                "learning_rate": learning_rate}, filepath)


def compute_validation_loss(model, criterion, valset, batch_size, n_gpus, apply_ctc):   # train.py:142-202
    model.eval()
    with torch.no_grad():
        collate_fn = DataCollate(n_frames_per_step=1, use_attn_prior=valset.use_attn_prior)
        val_loader = DataLoader(valset, sampler=None, num_workers=1, shuffle=False, batch_size=batch_size, pin_memory=False,
                                collate_fn=collate_fn)
        val_loss = 0.0
        for i, batch in enumerate(val_loader):
            mel, spk_ids, txt, in_lens, out_lens, gate_target, attn_prior = batch
            mel, spk_ids, txt = mel.cuda(), spk_ids.cuda(), txt.cuda()
            in_lens, out_lens = in_lens.cuda(), out_lens.cuda()
            gate_target = gate_target.cuda()
            attn_prior = attn_prior.cuda() if attn_prior is not None else None
            z, log_s_list, gate_pred, attn, attn_logprob, mean, log_var, prob = model(mel, spk_ids, txt, in_lens, out_lens, attn_prior)
            loss_nll, loss_gate, loss_ctc = criterion((z, log_s_list, gate_pred, attn, attn_logprob, mean, log_var, prob),
                                                      gate_target, in_lens, out_lens, is_validation=True)
            loss = loss_nll + loss_gate
            if apply_ctc:
                loss += loss_ctc * criterion.ctc_loss_weight
            val_loss += loss.item()
        val_loss = val_loss / len(val_loader)
    model.train()
    return val_loss


def train(config, max_iterations, text_frontend=grapheme_frontend):                # train.py:205-377, n_gpus = 1, rank = 0
    tc, data_config, model_config = config["train_config"], config["data_config"], config["model_config"]
    fp16_run, use_ctc_loss = bool(tc["fp16_run"]), bool(tc["use_ctc_loss"])
    torch.manual_seed(tc["seed"])
    torch.cuda.manual_seed(tc["seed"])
    criterion = FlowtronLoss(tc["sigma"], bool(model_config["n_components"]), tc["gate_loss"], use_ctc_loss, tc["ctc_loss_weight"],
                             tc["blank_logprob"])
    model = Flowtron(**model_config).cuda()
    optimizer = RAdam(model.parameters(), lr=tc["learning_rate"], weight_decay=tc["weight_decay"])
    iteration = 0
    if tc["checkpoint_path"] != "":
        model, optimizer, iteration = load_checkpoint(tc["checkpoint_path"], model, optimizer, tc["ignore_layers"])
        iteration += 1
    scaler = amp.GradScaler(enabled=fp16_run)
    train_loader, valset, collate_fn = prepare_dataloaders(data_config, 1, tc["batch_size"], text_frontend)
    os.makedirs(tc["output_directory"], exist_ok=True)
    for param_group in optimizer.param_groups:
        param_group["lr"] = tc["learning_rate"]
    model.train()
    apply_ctc = False
    losses, val_losses, checkpoints = {}, {}, []
    for epoch in range(tc["epochs"]):
        for batch in train_loader:
            model.zero_grad()
            mel, spk_ids, txt, in_lens, out_lens, gate_target, attn_prior = batch
            mel, spk_ids, txt = mel.cuda(), spk_ids.cuda(), txt.cuda()
            in_lens, out_lens = in_lens.cuda(), out_lens.cuda()
            gate_target = gate_target.cuda()
            attn_prior = attn_prior.cuda() if attn_prior is not None else None
            if use_ctc_loss and iteration >= tc["ctc_loss_start_iter"]:
                apply_ctc = True
            with amp.autocast(enabled=fp16_run):
                z, log_s_list, gate_pred, attn, attn_logprob, mean, log_var, prob = model(mel, spk_ids, txt, in_lens, out_lens, attn_prior)
                loss_nll, loss_gate, loss_ctc = criterion((z, log_s_list, gate_pred, attn, attn_logprob, mean, log_var, prob),
                                                          gate_target, in_lens, out_lens, is_validation=False)
                loss = loss_nll + loss_gate
                if apply_ctc:
                    loss += loss_ctc * criterion.ctc_loss_weight
            losses[iteration] = loss.item()
            scaler.scale(loss).backward()
            if tc["grad_clip_val"] > 0:
                scaler.unscale_(optimizer)
                torch.nn.utils.clip_grad_norm_(model.parameters(), tc["grad_clip_val"])
            scaler.step(optimizer)
            scaler.update()
            if iteration % tc["iters_per_checkpoint"] == 0:
                val_losses[iteration] = compute_validation_loss(model, criterion, valset, tc["batch_size"], 1, apply_ctc)
                checkpoint_path = "{}/model_{}".format(tc["output_directory"], iteration)
                save_checkpoint(model, optimizer, tc["learning_rate"], iteration, checkpoint_path, model_config)
                checkpoints.append(checkpoint_path)
            iteration += 1
            if iteration >= max_iterations:
                return dict(losses=losses, val_losses=val_losses, checkpoints=checkpoints, model=model, optimizer=optimizer,
                            scale=scaler.get_scale() if fp16_run else 1.0)
    return dict(losses=losses, val_losses=val_losses, checkpoints=checkpoints, model=model, optimizer=optimizer, scale=1.0)


def infer(config, flowtron_path, text, speaker_id, n_frames, sigma, gate_threshold, seed, text_frontend=grapheme_frontend):
    """inference.py:40-71 without the vocoder (WaveGlow is an empty submodule in the reference checkout)."""
    data_config, model_config = config["data_config"], config["model_config"]
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    model = Flowtron(**model_config).cuda()
    ck = torch.load(flowtron_path, map_location="cpu")
    state_dict = ck["state_dict"] if "state_dict" in ck else ck["model"].state_dict()
    model.load_state_dict(state_dict)
    model.eval()
    ignore_keys = ["training_files", "validation_files"]
    trainset = Data(data_config["training_files"], text_frontend=text_frontend,
                    **dict((k, v) for k, v in data_config.items() if k not in ignore_keys))
    speaker_vecs = trainset.get_speaker_id(speaker_id).cuda()
    text = trainset.get_text(text).cuda()
    speaker_vecs = speaker_vecs[None]
    text = text[None]
    with torch.no_grad():
        residual = torch.randn(1, 80, n_frames, device="cuda") * sigma
        mels, attentions = model.infer(residual, speaker_vecs, text, gate_threshold=gate_threshold)
    return mels, attentions
