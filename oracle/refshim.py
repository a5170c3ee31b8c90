"""Import the REAL reference (`/root/reference/flowtron.py`) with the two
monkey-patches SURVEY 8c documents -- build-container only, never on the GPU
box. Used solely by tests/golden/make_golden.py and by oracle-vs-reference
checks that skip when /root/reference is absent. TEST INFRASTRUCTURE ONLY."""
import importlib.util
import os
import sys

import torch

REF_DIR = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REF_DIR, "flowtron.py"))


def load():
    """Returns the reference `flowtron` module object under the private name
    `_ref_flowtron` (so it never shadows this repo's own top-level flowtron.py)."""
    if "_ref_flowtron" in sys.modules:
        return sys.modules["_ref_flowtron"]
    spec = importlib.util.spec_from_file_location("_ref_flowtron", os.path.join(REF_DIR, "flowtron.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_ref_flowtron"] = mod
    spec.loader.exec_module(mod)

    def get_mask_from_lengths(lengths):           # flowtron.py:47-48 hard-codes torch.cuda.LongTensor
        max_len = int(torch.max(lengths).item())
        ids = torch.arange(0, max_len, device=lengths.device)
        return (ids < lengths.unsqueeze(1)).bool()

    mod.get_mask_from_lengths = get_mask_from_lengths
    if not torch.cuda.is_available():             # flowtron.py:785 torch.cuda.FloatTensor
        torch.cuda.FloatTensor = torch.FloatTensor
    return mod
