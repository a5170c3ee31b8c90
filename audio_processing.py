"""Drop-in for the reference's `audio_processing.py` (data.py:27 does `from audio_processing import
TacotronSTFT`): same class names and the `mel_spectrogram(y)` contract, HIP kernel underneath.
iSTFT / Griffin-Lim (audio_processing.py:7-75, 237-270) are never called by the train/inference path and are
out of scope."""
from flowtron_amd.audio import (STFT, TacotronSTFT, dynamic_range_compression,  # noqa: F401
                                dynamic_range_decompression)

for _cls in (STFT, TacotronSTFT):
    _cls.__module__ = "audio_processing"
