"""The LibriTTS-shaped full-width case (BASELINE configs[2] / configs[4]: config.json:50-65 with n_speakers = 123) shared by
tests/golden/make_golden_r2.py (which runs the REAL reference on it), tests/test_oracle_golden.py (oracle vs that golden, CPU)
and tests/test_gpu_bench_path.py (HIP path vs oracle and golden): H = 1024, 2 flows, B = 4 ragged utterances, texts of
237 / 231 / 129 / 5 symbols (two 128-column score tiles, a tile edge, the minimum; 475 CTC states), speaker ids spread over the
123-row embedding, attention prior + CTC on."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED_WEIGHTS = 41
OUT_LENS, IN_LENS, SPEAKERS = [320, 301, 214, 160], [237, 231, 129, 5], [122, 0, 57, 101]


def make():
    """-> (cfg, state_dict, batch dict with mel / speaker_ids / text / in_lens / out_lens / gate_target / attn_prior)"""
    import bench
    from oracle import synth
    cfg = dict(bench.MODEL_CONFIG, n_speakers=123)
    bb = bench.synth_batch(4, 99, t_max=320, l_max=237, l_min=5, n_speakers=123, chars_per_frame=1 / 1.3)
    bb["speaker_ids"] = torch.tensor(SPEAKERS)
    bb["out_lens"] = torch.tensor(OUT_LENS)
    bb["in_lens"] = torch.tensor(IN_LENS)
    for i in range(4):
        bb["mel"][i, :, OUT_LENS[i]:] = 0
        bb["text"][i, IN_LENS[i]:] = 0
        bb["gate"][i] = 0
        bb["gate"][i, OUT_LENS[i] - 1:] = 1
    T, Lk = bb["mel"].shape[2], bb["text"].shape[1]
    assert (T, Lk) == (320, 237)
    prior = bench.beta_binomial_prior_batch(bb["in_lens"], bb["out_lens"], T, Lk)
    batch = dict(mel=bb["mel"], speaker_ids=bb["speaker_ids"], text=bb["text"], in_lens=bb["in_lens"], out_lens=bb["out_lens"],
                 gate_target=bb["gate"], attn_prior=prior)
    return cfg, synth.make_state_dict(cfg, seed=SEED_WEIGHTS), batch
