"""Drop-in for the reference's `data.py` (train.py:27 `from data import Data, DataCollate`; inference.py:31): same class
names and constructor arguments, the 7-tuple wire format of `DataCollate` -- with the mel spectrogram (audio_processing.py:117-134)
and the beta-binomial attention prior (data.py:31-41) computed on the MI355X when the training loop calls `.cuda()` on their
slots, instead of per item on the host inside the single DataLoader worker (flowtron_amd/data.py)."""
from flowtron_amd.data import (Data, DataCollate, DeferredMel, DeferredPrior, LengthBucketBatchSampler,  # noqa: F401
                               load_filepaths_and_text, load_wav_to_torch)

for _cls in (Data, DataCollate):
    _cls.__module__ = "data"
