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

from typing import Optional, Sequence

import torch

from . import ops


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
