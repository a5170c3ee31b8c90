"""The CPU oracle over a LARGE batch, a few utterances at a time (test infrastructure).

Utterances are independent through forward, loss and backward (instance norm is per sample, no batch statistics anywhere --
SURVEY 8e), and every loss term is a sum over utterances divided by a batch-level count:
    nll, gate : sum over valid frames / (number of valid frames of the WHOLE batch [* n_mel])      flowtron.py:206-243
    ctc       : mean over samples (and flows) of the per-sample CTC                                 flowtron.py:162-182, 245-274
so the oracle's losses and gradients for the whole batch are the weighted sums of those of sub-batches (weights n_sub / n and
B_sub / B), each evaluated at its own padded size.  At BASELINE configs[1]'s own shape (B 32, T 862, L 157) the one-shot oracle
would keep ~50 GB of autograd state (the B x T x L x A tanh tensors); four utterances at a time stay under 8 GB.
"""
import torch

from oracle import flowtron_oracle as O


def forward_backward(cfg, sd, batch, prior, chunk=4, ctc_weight=0.01, blank_logprob=-8):
    """batch: mel [B,M,T], speaker_ids, text [B,L], in_lens, out_lens, gate_target [B,T] (sorted by in_lens, descending);
    prior [B,T,L] or None.  Returns ((nll, gate, ctc) floats, {name: grad}) of loss = nll + gate + ctc_weight * ctc."""
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    B = batch["mel"].shape[0]
    n_total = float(batch["out_lens"].sum())
    tot = [0.0, 0.0, 0.0]
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        out_lens, in_lens = batch["out_lens"][sl], batch["in_lens"][sl]
        T, Lk = int(out_lens.max()), int(in_lens.max())
        pr = None if prior is None else prior[sl, :T, :Lk].contiguous()
        out = O.forward(sdg, cfg, batch["mel"][sl][:, :, :T].contiguous(), batch["speaker_ids"][sl], batch["text"][sl][:, :Lk].contiguous(),
                        in_lens, out_lens, pr)
        nll, gl, ctc = O.loss(out, batch["gate_target"][sl][:, :T].contiguous(), in_lens, out_lens, 1.0, True, True, blank_logprob)
        wf, wb = float(out_lens.sum()) / n_total, float(sl.stop - sl.start) / B
        (wf * nll + wf * gl + ctc_weight * wb * ctc).sum().backward()
        tot[0] += wf * float(nll.detach())
        tot[1] += wf * float(gl.detach())
        tot[2] += wb * float(ctc.detach())
        del out, nll, gl, ctc
    return tuple(tot), {k: v.grad for k, v in sdg.items()}
