"""Collate + host->device staging: mirror of the reference's `DataCollate` (data.py:191-246) with the two MI355X-side
changes SURVEY 8f ranks 1 and 4 call for:

  * the beta-binomial attention prior (data.py:31-41, 111-141: ~0.45 s per utterance in scipy, computed per item in a
    single DataLoader worker) is evaluated for the WHOLE batch by one HIP kernel (`ft_beta_binomial_prior`) after the
    lengths are on the device;
  * the padded batch is assembled in PINNED host memory and copied with non_blocking H2D copies on a side stream, so the
    next batch's transfer overlaps the current step (the reference does 7 synchronous `.cuda()` calls from pageable
    memory, train.py:285-288).

Wire format is unchanged: (mel_padded [B,M,T], speaker_ids [B], text_padded [B,L], input_lengths [B],
output_lengths [B], gate_padded [B,T], attn_prior_padded [B,T,L] | None), sorted by text length descending.
"""
from __future__ import annotations

import os
import random
import re
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops


class DeferredMel:
    """The `mel` slot of a collated batch when the items carry AUDIO instead of spectrograms (drop-in `Data` below).
    The reference computes every mel on the host, one item at a time, inside the single DataLoader worker
    (data.py:149-155, train.py:77); a forked worker cannot touch the GPU, so the worker ships the padded audio and the
    spectrogram is computed by the HIP front end (ft_stft_mel) when the training loop calls `.cuda()` on the slot
    (train.py:286) -- the one call the reference makes on it.  Result: [B, n_mel, T_max] fp32 on the device, zero beyond
    each utterance's frames, exactly what DataCollate's zero padding produces (data.py:215-229)."""

    def __init__(self, audio, n_samples, stft_args, max_t=None):
        # max_t: the padded frame count DataCollate uses for every slot of the batch (rounded up to a multiple of
        # n_frames_per_step, data.py:213-216); None = the longest utterance's own frame count
        self.audio, self.n_samples, self.stft_args, self.max_t = audio, n_samples, dict(stft_args), max_t

    def __len__(self):
        return self.audio.shape[0]

    def cuda(self, device=None, non_blocking=False):
        from .audio import TacotronSTFT
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        key = (tuple(sorted(self.stft_args.items())), str(dev))
        stft = _STFT_CACHE.get(key)
        if stft is None:
            stft = _STFT_CACHE[key] = TacotronSTFT(**self.stft_args).to(dev)
        # H2D through a cached PINNED staging buffer: the padded audio arrives from the DataLoader worker in a shared-memory
        # mapping, and a direct copy from such pageable memory makes the runtime register (pin + map) the user pages on every
        # batch -- measured at ~10 s per batch in a long-lived process with a large address space (a -m gpu suite run: 170 s
        # per training-loop test against 13 s in a fresh process).  A host memcpy into pinned memory + a DMA costs nothing.
        audio = _pinned_stage(self.audio).to(dev, non_blocking=False)
        hop = stft.stft_fn.hop_length
        frames = [int(n) // hop + 1 for n in self.n_samples.tolist()]
        max_t = max(frames) if self.max_t is None else max(int(self.max_t), max(frames))
        if (stft.stft_fn.fast_path() and stft.n_mel_channels <= 128
                and int(self.n_samples.min()) > stft.stft_fn.filter_length // 2):
            # ONE launch for the batch: every utterance reflected about its own last sample, zeros behind its last frame.
            # Same preconditions as TacotronSTFT.mel_spectrogram's fast kernel: more than 128 mel channels, or an
            # utterance no longer than the reflection pad (the reference's reflect pad raises on those), take the
            # per-utterance loop below, whose mel_spectrogram falls back to the general front end (ft_stft_mel)
            return stft.mel_spectrogram_ragged(audio, self.n_samples.to(dev, dtype=torch.int32), max_t)
        mel = torch.zeros(len(frames), stft.n_mel_channels, max_t, device=dev, dtype=torch.float32)
        for i, (n, t) in enumerate(zip(self.n_samples.tolist(), frames)):       # reflect padding depends on each utterance's end
            mel[i, :, :t] = stft.mel_spectrogram(audio[i:i + 1, :int(n)])[0]
        return mel

    def to(self, device, non_blocking=False):
        return self.cuda(device, non_blocking)


class DeferredPrior:
    """The `attn_prior` slot: the beta-binomial prior (data.py:31-41, 111-141; ~0.45 s per utterance in scipy) is evaluated
    for the whole batch by ft_beta_binomial_prior when the loop calls `.cuda()` (train.py:288)."""

    def __init__(self, in_lens, out_lens, max_t, max_in, scaling, threshold):
        self.in_lens, self.out_lens, self.max_t, self.max_in = in_lens, out_lens, int(max_t), int(max_in)
        self.scaling, self.threshold = float(scaling), float(threshold)

    def cuda(self, device=None, non_blocking=False):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pr = ops.beta_binomial_prior(self.in_lens.to(dev), self.out_lens.to(dev), self.max_t, self.max_in, self.scaling)
        if self.threshold > 0:
            pr = pr.masked_fill(pr < self.threshold, 0.0)                        # data.py:136-138
        return pr

    def to(self, device, non_blocking=False):
        return self.cuda(device, non_blocking)


_STFT_CACHE = {}
_PIN = {}


def _pinned_stage(t):
    """`t` copied into a grow-only pinned host buffer (one per dtype); the caller must finish its H2D copy before the next
    call (DeferredMel.cuda copies synchronously)."""
    n = t.numel()
    buf = _PIN.get(t.dtype)
    if buf is None or buf.numel() < n:
        buf = _PIN[t.dtype] = torch.empty(max(n, 1 << 20), dtype=t.dtype, pin_memory=True)
    v = buf[:n].view(t.shape)
    v.copy_(t)
    return v


class AudioItem:
    """What `Data.__getitem__` puts in the mel slot of an item: normalised audio + the frame count DataCollate needs."""
    __slots__ = ("audio", "n_frames", "stft_args", "prior_args")

    def __init__(self, audio, n_frames, stft_args, prior_args=None):
        self.audio, self.n_frames, self.stft_args, self.prior_args = audio, n_frames, stft_args, prior_args

    def size(self, dim):            # DataCollate reads x[0].size(0) = n_mel and x[0].size(1) = frames (data.py:211-212)
        return self.stft_args["n_mel_channels"] if dim == 0 else self.n_frames


def load_filepaths_and_text(filelist, split="|"):
    """data.py:44-50."""
    if isinstance(filelist, str):
        with open(filelist, encoding="utf-8") as f:
            return [line.strip().split(split) for line in f]
    return filelist


def load_wav_to_torch(full_path):
    """data.py:53-56."""
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.from_numpy(np.ascontiguousarray(data)).float(), sampling_rate


class Data(torch.utils.data.Dataset):
    """Drop-in for the reference dataset (data.py:59-188: same constructor arguments = the `data_config` keys of config.json,
    same `get_text` / `get_speaker_id` / `speaker_ids` surface that train.py:69-71 and inference.py:58-62 use).
    MI355X-side change: items carry normalised AUDIO (AudioItem); the mel spectrogram and the attention prior are computed on
    the device, per batch, after collate (DeferredMel / DeferredPrior) -- the DataLoader worker only reads files and runs
    the text front end.  The text front end itself (cleaners, CMUdict, ARPAbet; data.py:28, 164-171) is outside the hot path:
    the reference's `text` package is used when it is importable, otherwise `text_frontend` must be given."""

    def __init__(self, filelist_path, filter_length, hop_length, win_length, sampling_rate, mel_fmin, mel_fmax, max_wav_value,
                 p_arpabet, cmudict_path, text_cleaners, speaker_ids=None, use_attn_prior=False, attn_prior_threshold=1e-4,
                 prior_cache_path="", betab_scaling_factor=1.0, randomize=True, keep_ambiguous=False, seed=1234,
                 text_frontend=None):
        self.max_wav_value = max_wav_value
        self.audiopaths_and_text = load_filepaths_and_text(filelist_path)
        self.use_attn_prior = use_attn_prior
        self.betab_scaling_factor = betab_scaling_factor
        self.attn_prior_threshold = attn_prior_threshold
        self.keep_ambiguous = keep_ambiguous
        if speaker_ids is None or speaker_ids == "":
            self.speaker_ids = self.create_speaker_lookup_table(self.audiopaths_and_text)
        else:
            self.speaker_ids = speaker_ids
        self.stft_args = dict(filter_length=filter_length, hop_length=hop_length, win_length=win_length, n_mel_channels=80,
                              sampling_rate=sampling_rate, mel_fmin=mel_fmin, mel_fmax=mel_fmax)
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.text_cleaners = text_cleaners
        self.p_arpabet = p_arpabet
        self.text_frontend = text_frontend
        self._ref_text = None
        if text_frontend is None:
            try:                                   # the reference's own front end (text/__init__.py, text/cmudict.py)
                import text as _text
                self._ref_text = _text
                self.cmudict = _text.cmudict.CMUDict(cmudict_path, keep_ambiguous=keep_ambiguous)
            except Exception as e:                 # not importable here: a front end must be supplied by the caller
                raise ImportError("flowtron_amd.data.Data needs the reference's `text` package on sys.path (cwd = the reference "
                                  "root, text/__init__.py:120 opens relative paths) or an explicit text_frontend=callable: %r" % (e,))
        self.prior_cache_path = prior_cache_path      # accepted for config.json compatibility; the device prior needs no cache
        random.seed(seed)
        if randomize:
            random.shuffle(self.audiopaths_and_text)

    def create_speaker_lookup_table(self, audiopaths_and_text):
        speaker_ids = np.sort(np.unique([x[2] for x in audiopaths_and_text]))
        d = {int(speaker_ids[i]): i for i in range(len(speaker_ids))}
        print("Number of speakers :", len(d))
        return d

    def get_speaker_id(self, speaker_id):
        return torch.LongTensor([self.speaker_ids[int(speaker_id)]])

    def get_text(self, text):
        if self.text_frontend is not None:
            return torch.LongTensor(list(self.text_frontend(text)))
        t = self._ref_text
        text = t._clean_text(text, self.text_cleaners)
        words = re.findall(r"\S*\{.*?\}\S*|\S+", text)
        text = " ".join([t.get_arpabet(word, self.cmudict) if random.random() < self.p_arpabet else word for word in words])
        return torch.LongTensor(t.text_to_sequence(text))

    def get_mel(self, audio):
        """data.py:149-155 (used by the reference's mel-dump tool): here on the device."""
        from .audio import TacotronSTFT
        dev = torch.device("cuda", torch.cuda.current_device())
        stft = TacotronSTFT(**self.stft_args).to(dev)
        return stft.mel_spectrogram((audio / self.max_wav_value).unsqueeze(0).to(dev))[0]

    def __getitem__(self, index):
        audiopath, text, speaker_id = self.audiopaths_and_text[index]
        audio, sampling_rate = load_wav_to_torch(audiopath)
        if sampling_rate != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sampling_rate, self.sampling_rate))
        audio = audio / self.max_wav_value
        item = AudioItem(audio, audio.numel() // self.hop_length + 1, self.stft_args,
                         (self.betab_scaling_factor, self.attn_prior_threshold))     # train.py:69 builds DataCollate without them
        return (item, self.get_speaker_id(speaker_id), self.get_text(text), None)

    def __len__(self):
        return len(self.audiopaths_and_text)


class DataCollate:
    """Zero-pads model inputs and targets (data.py:191-246).  Items are `(mel [M,T_i], speaker_id, text [L_i], prior|None)`.
    With `device=None` the behaviour (and the returned CPU tensors) equals the reference's.  With a device, the tuple is
    returned as device tensors (async copies from pinned buffers); if `use_attn_prior` and the items carry no prior, the
    prior is computed on the device."""

    def __init__(self, n_frames_per_step=1, use_attn_prior=False, device: Optional[torch.device] = None,
                 betab_scaling_factor: float = 1.0, attn_prior_threshold: float = 0.0):
        self.n_frames_per_step = n_frames_per_step
        self.use_attn_prior = use_attn_prior
        self.device = torch.device(device) if device is not None else None
        self.betab_scaling_factor = betab_scaling_factor
        self.attn_prior_threshold = attn_prior_threshold
        self._copy_stream = None

    def _host(self, *shape, dtype):
        pin = self.device is not None and self.device.type == "cuda"
        return torch.zeros(*shape, dtype=dtype, pin_memory=pin)

    def __call__(self, batch: Sequence):
        B = len(batch)
        if isinstance(batch[0][0], AudioItem):
            return self._collate_audio(batch)
        input_lengths, order = torch.sort(torch.tensor([len(x[2]) for x in batch], dtype=torch.long), dim=0, descending=True)
        max_in = int(input_lengths[0])
        text_padded = self._host(B, max_in, dtype=torch.long)
        n_mel = batch[0][0].size(0)
        max_t = max(x[0].size(1) for x in batch)
        if max_t % self.n_frames_per_step != 0:
            max_t += self.n_frames_per_step - max_t % self.n_frames_per_step
        mel_padded = self._host(B, n_mel, max_t, dtype=torch.float32)
        gate_padded = self._host(B, max_t, dtype=torch.float32)
        output_lengths = self._host(B, dtype=torch.long)
        speaker_ids = self._host(B, dtype=torch.long)
        have_item_prior = self.use_attn_prior and all(len(x) > 3 and x[3] is not None for x in batch)
        prior_padded = self._host(B, max_t, max_in, dtype=torch.float32) if have_item_prior else None
        for i, j in enumerate(order.tolist()):
            mel, spk, text = batch[j][0], batch[j][1], batch[j][2]
            text_padded[i, :text.size(0)] = text
            mel_padded[i, :, :mel.size(1)] = mel
            gate_padded[i, mel.size(1) - 1:] = 1
            output_lengths[i] = mel.size(1)
            speaker_ids[i] = int(spk)
            if have_item_prior:
                p = batch[j][3]
                prior_padded[i, :p.size(0), :p.size(1)] = p
        if self.device is None:
            if self.use_attn_prior and not have_item_prior:
                raise ValueError("use_attn_prior on the host path needs per-item priors (the device path computes them)")
            return (mel_padded, speaker_ids, text_padded, input_lengths, output_lengths, gate_padded, prior_padded)
        return self.to_device(mel_padded, speaker_ids, text_padded, input_lengths, output_lengths, gate_padded, prior_padded,
                              max_t, max_in)

    def _collate_audio(self, batch):
        """items from the drop-in `Data`: same 7-tuple, with the mel (and prior) slots deferred to the device."""
        B = len(batch)
        input_lengths, order = torch.sort(torch.tensor([len(x[2]) for x in batch], dtype=torch.long), dim=0, descending=True)
        max_in = int(input_lengths[0])
        text_padded = torch.zeros(B, max_in, dtype=torch.long)
        max_n = max(x[0].audio.numel() for x in batch)
        max_t = max(x[0].n_frames for x in batch)
        if max_t % self.n_frames_per_step != 0:
            max_t += self.n_frames_per_step - max_t % self.n_frames_per_step
        audio = torch.zeros(B, max_n, dtype=torch.float32)
        n_samples = torch.zeros(B, dtype=torch.long)
        gate_padded = torch.zeros(B, max_t, dtype=torch.float32)
        output_lengths = torch.zeros(B, dtype=torch.long)
        speaker_ids = torch.zeros(B, dtype=torch.long)
        for i, j in enumerate(order.tolist()):
            it, spk, text = batch[j][0], batch[j][1], batch[j][2]
            text_padded[i, :text.size(0)] = text
            audio[i, :it.audio.numel()] = it.audio
            n_samples[i] = it.audio.numel()
            gate_padded[i, it.n_frames - 1:] = 1
            output_lengths[i] = it.n_frames
            speaker_ids[i] = int(spk)
        mel = DeferredMel(audio, n_samples, batch[0][0].stft_args, max_t=max_t)
        scaling, thr = batch[0][0].prior_args or (self.betab_scaling_factor, self.attn_prior_threshold)
        prior = DeferredPrior(input_lengths, output_lengths, max_t, max_in, scaling, thr) if self.use_attn_prior else None
        return (mel, speaker_ids, text_padded, input_lengths, output_lengths, gate_padded, prior)

    def to_device(self, mel, spk, text, in_lens, out_lens, gate, prior, max_t, max_in):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("DataCollate(device=...) stages to an MI355X; got %s" % dev)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(self._copy_stream):
            out = [t.to(dev, non_blocking=True) if t is not None else None for t in (mel, spk, text, in_lens, out_lens, gate, prior)]
        cur.wait_stream(self._copy_stream)
        for t in out:
            if t is not None:
                t.record_stream(cur)
        if self.use_attn_prior and out[6] is None:
            pr = ops.beta_binomial_prior(out[3], out[4], max_t, max_in, self.betab_scaling_factor)
            if self.attn_prior_threshold > 0:
                pr = pr.masked_fill(pr < self.attn_prior_threshold, 0.0)      # data.py:136-138
            out[6] = pr
        return tuple(out)


class LengthBucketBatchSampler(torch.utils.data.Sampler):
    """Batches of similar mel length.  The reference sorts a batch by TEXT length only and draws batches
    uniformly at random (data.py:200-202, train.py:74-80), so B*T_max carries ~30 % padding at LJSpeech lengths -- and every
    recurrence of the hot path runs T_max launches whatever the short utterances need.  This sampler keeps the epoch a
    random permutation, but forms batches inside shuffled buckets of `bucket_batches * batch_size * world_size` utterances
    sorted by length, then shuffles the batch order and deals batches to ranks round-robin (every rank gets the same number
    of batches, equal-cost batches in the same step).  Opt-in: `DataLoader(dataset, batch_sampler=LengthBucketBatchSampler(
    lengths, batch_size, rank=rank, world_size=n), collate_fn=DataCollate(...))` instead of `sampler=DistributedSampler`.
    `drop_last` semantics as train.py:79 (incomplete batches / uneven tails are dropped)."""

    def __init__(self, lengths: Sequence[int], batch_size: int, bucket_batches: int = 16, rank: int = 0, world_size: int = 1,
                 shuffle: bool = True, seed: int = 1234):
        self.lengths = torch.as_tensor(list(lengths), dtype=torch.int64)
        assert batch_size >= 1 and bucket_batches >= 1 and 0 <= rank < world_size
        self.batch_size, self.bucket_batches = int(batch_size), int(bucket_batches)
        self.rank, self.world_size, self.shuffle, self.seed, self.epoch = int(rank), int(world_size), bool(shuffle), int(seed), 0
        self.n_batches = (len(self.lengths) // self.batch_size) // self.world_size      # per rank

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n_batches

    def _all_batches(self):
        n = len(self.lengths)
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
        bucket = self.batch_size * self.bucket_batches * self.world_size
        batches = []
        for s in range(0, n, bucket):
            idx = order[s:s + bucket]
            idx = idx[torch.argsort(self.lengths[idx], descending=True, stable=True)]
            for b in range(0, len(idx) - self.batch_size + 1, self.batch_size):
                batches.append(idx[b:b + self.batch_size])
        if self.shuffle:                                  # keep groups of world_size consecutive (similar-length) batches
            groups = [batches[i:i + self.world_size] for i in range(0, len(batches) - self.world_size + 1, self.world_size)]
            perm = torch.randperm(len(groups), generator=g).tolist()
            batches = [bt for k in perm for bt in groups[k]]
        return batches[:self.n_batches * self.world_size]

    def __iter__(self):
        for bt in self._all_batches()[self.rank::self.world_size]:
            yield bt.tolist()
