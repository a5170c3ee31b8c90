"""Builds flowtron_amd/libflowtron_hip.so (gfx950 only) from csrc/*.hip with hipcc.

In-tree build: the .so sits next to this file so it travels with the repo snapshot to
the GPU box (it is git-ignored, not gpurun-ignored).  `python -m flowtron_amd.build`.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libflowtron_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("FT_EXTRA_HIPCC", "").split()


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libflowtron_hip.so cannot be built")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


# the operand-typed files are compiled twice: bf16 operands (default) and fp16 operands (-DFT_OPFMT=1, entries suffixed _f16)
OP16_SOURCES = ("gemm.hip", "gemm_bf16.hip", "lstm.hip", "lstm2.hip", "lstm_persist.hip", "lstm_roles.hip", "bilstm_persist.hip", "cumm_fused.hip")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def units():
    """(source, extra flags, object suffix) per compilation."""
    u = [(s, [], "") for s in sources()]
    u += [(os.path.join(CSRC, f), ["-DFT_OPFMT=1"], "_f16") for f in OP16_SOURCES]
    return u


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "flowtron_hip.h"))
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    dg = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dg:
        return LIB
    hipcc = _hipcc()
    hd = _digest(hdrs)

    def compile_one(unit):
        src, extra, suffix = unit
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + suffix + ".o")
        tag = obj + ".tag"
        d = _digest([src]) + hd + " ".join(extra)
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == d:
            return obj
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        with open(tag, "w") as f:
            f.write(d)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, units()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dg)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
