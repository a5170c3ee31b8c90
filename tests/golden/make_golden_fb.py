"""Golden mel filterbank for `audio_processing.py:104-107` (`librosa_mel_fn(22050, 1024, 80, 0.0, 8000.0)`).

librosa itself is not installed in the build container (un-vendored dependency of the reference, requirements.txt:4), so the
constants cannot come from the reference's own call.  The closest independent, published implementation that IS here is
Hugging Face `transformers.audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney")` (transformers 5.15.0, wheel
from the offline wheelhouse), whose documented contract is equality with `librosa.filters.mel` (it is what the Whisper / CLAP
feature extractors replaced librosa with, and their test-suites hold librosa-generated spectrogram fixtures).  This script
writes its output as the fixture that pins BOTH restatements in this repo (oracle.mel_filterbank and
flowtron_amd.audio.slaney_mel_filterbank).  It is third-party-vs-third-party, not the reference's own librosa call: DESIGN.md
says so.

    python tests/golden/make_golden_fb.py        # -> tests/golden/mel_fb_hf_slaney.npz
"""
import os

import numpy as np


def main():
    import transformers
    from transformers.audio_utils import mel_filter_bank
    fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                         sampling_rate=22050, norm="slaney", mel_scale="slaney")           # [513, 80] float64
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_fb_hf_slaney.npz")
    np.savez_compressed(out, fb=np.ascontiguousarray(fb.T), source="transformers.audio_utils.mel_filter_bank",
                        version=transformers.__version__,
                        args="num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0, "
                             "sampling_rate=22050, norm='slaney', mel_scale='slaney'")
    print(out, fb.T.shape, fb.dtype, float(fb.max()))


if __name__ == "__main__":
    main()
