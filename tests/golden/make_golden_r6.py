"""Round 6 golden vectors from the REAL reference (/root/reference), run once in the build container:
    python tests/golden/make_golden_r6.py
  small_dummy_spk.pt   the 2-flow small model (3 speakers, distinct speaker ids in the batch) with dummy_speaker_embedding=True
                       (flowtron.py:872-873: every utterance reads speaker 0's embedding): every forward output, the three
                       losses, all gradients (the speaker embedding's gradient lands in row 0 only), infer mel
Inputs and weights are rebuilt from seeds by oracle/synth.py, like the other fixtures."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import refshim, synth  # noqa: E402
import make_golden as G1  # noqa: E402


def main():
    assert refshim.available(), "needs /root/reference"
    torch.set_num_threads(8)
    R = refshim.load()
    cfg = dict(synth.SMALL_MODEL_CONFIG, dummy_speaker_embedding=True)
    torch.save(G1.run_model_case(R, cfg, [21, 16, 19], [8, 7, 5], True, 9, True, 12), os.path.join(HERE, "small_dummy_spk.pt"))
    g = torch.load(os.path.join(HERE, "small_dummy_spk.pt"), weights_only=False)
    sg = g["grads"]["speaker_embedding.weight"]
    print("speaker ids of the batch:", synth.make_batch(cfg, [21, 16, 19], [8, 7, 5], seed=9)["speaker_ids"].reshape(-1).tolist(),
          "| |grad| per speaker row:", [round(float(sg[i].norm()), 6) for i in range(sg.shape[0])])


if __name__ == "__main__":
    main()
