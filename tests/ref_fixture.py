"""Synthetic LJSpeech-shaped data set on disk (wav files + a train.py-format filelist + a config.json) for the tests that
drive the reference's call sequence.  TEST INFRASTRUCTURE."""
import json
import os

import numpy as np

SENTENCES = ["the quick brown fox jumps over the lazy dog", "printing in the only sense with which we are concerned",
             "a short one", "flowtron is an autoregressive flow based generative network for text to speech synthesis",
             "many years later he remembered that distant afternoon", "she sells sea shells by the sea shore"]


def make_dataset(root, n_utt=12, n_speakers=1, sr=22050, seed=0, min_s=0.35, max_s=0.9):
    """writes n_utt int16 wavs (0.35-0.9 s of mixed sinusoids) and returns (train_filelist, val_filelist)."""
    from scipy.io.wavfile import write
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "wavs"), exist_ok=True)
    lines = []
    for i in range(n_utt):
        n = int(rs.uniform(min_s, max_s) * sr)
        t = np.arange(n) / sr
        y = sum(rs.uniform(0.2, 1.0) * np.sin(2 * np.pi * rs.uniform(80, 4000) * t + rs.uniform(0, 6.28)) for _ in range(6))
        y = 0.9 * y / np.abs(y).max()
        path = os.path.join(root, "wavs", "utt%03d.wav" % i)
        write(path, sr, (y * 32767.0).astype(np.int16))
        lines.append("%s|%s|%d" % (path, SENTENCES[i % len(SENTENCES)], i % n_speakers))
    tr, va = os.path.join(root, "train.txt"), os.path.join(root, "val.txt")
    open(tr, "w").write("\n".join(lines[: max(1, n_utt - 4)]) + "\n")
    open(va, "w").write("\n".join(lines[max(1, n_utt - 4):]) + "\n")
    return tr, va


def make_config(root, model_overrides=None, train_overrides=None, data_overrides=None, cmudict_path="data/cmudict_dictionary"):
    """a config.json with the reference's schema (config.json:1-67) pointing at the synthetic data set."""
    tr, va = make_dataset(root)
    cfg = {
        "train_config": {"output_directory": os.path.join(root, "out"), "epochs": 100, "optim_algo": "RAdam", "learning_rate": 1e-3,
                         "weight_decay": 1e-6, "grad_clip_val": 1, "sigma": 1.0, "iters_per_checkpoint": 2, "batch_size": 4,
                         "seed": 1234, "checkpoint_path": "", "ignore_layers": [], "finetune_layers": [],
                         "include_layers": ["speaker", "encoder", "embedding"], "warmstart_checkpoint_path": "",
                         "with_tensorboard": False, "fp16_run": False, "gate_loss": True, "use_ctc_loss": True,
                         "ctc_loss_weight": 0.01, "blank_logprob": -8, "ctc_loss_start_iter": 0},
        "data_config": {"training_files": tr, "validation_files": va, "text_cleaners": ["flowtron_cleaners"], "p_arpabet": 0.5,
                        "cmudict_path": cmudict_path, "sampling_rate": 22050, "filter_length": 1024, "hop_length": 256,
                        "win_length": 1024, "mel_fmin": 0.0, "mel_fmax": 8000.0, "max_wav_value": 32768.0, "use_attn_prior": True,
                        "attn_prior_threshold": 0.0, "prior_cache_path": "", "betab_scaling_factor": 1.0, "keep_ambiguous": False},
        "dist_config": {"dist_backend": "nccl", "dist_url": "tcp://localhost:54321"},
        "model_config": {"n_speakers": 1, "n_speaker_dim": 128, "n_text": 185, "n_text_dim": 512, "n_flows": 2, "n_mel_channels": 80,
                         "n_attn_channels": 640, "n_hidden": 1024, "n_lstm_layers": 2, "mel_encoder_n_hidden": 512,
                         "n_components": 0, "mean_scale": 0.0, "fixed_gaussian": True, "dummy_speaker_embedding": False,
                         "use_gate_layer": True, "use_cumm_attention": False},
    }
    cfg["model_config"].update(model_overrides or {})
    cfg["train_config"].update(train_overrides or {})
    cfg["data_config"].update(data_overrides or {})
    path = os.path.join(root, "config.json")
    json.dump(cfg, open(path, "w"), indent=1)
    return path, cfg
