"""STFT/mel front end on a LJSpeech-shaped batch (32 x 10 s at 22 050 Hz) -- a stand-alone workload for rocprofv3 passes over the front end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_processing
stft = audio_processing.TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0)
y = (torch.rand(32, 220500, device="cuda") * 2 - 1) * 0.9
for _ in range(5):
    mel = stft.mel_spectrogram(y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    mel = stft.mel_spectrogram(y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
frames = mel.shape[0] * mel.shape[2]
print("stft_mel: %.3f ms per batch, %d frames, %.1f Mframes/s, %.1f GB/s algorithmic (1344 B/frame)" % (ms, frames, frames / ms / 1e3, frames * 1344 / ms / 1e6))
