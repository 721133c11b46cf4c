"""Build libagile3d_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object loaded with ctypes (agile3d_amd/lib.py)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libagile3d_hip.so")
SOURCES = ["scene.hip", "radix.hip", "spconv.hip", "decoder.hip", "clicks.hip", "quantize.hip", "criterion.hip", "wgrad.hip", "bnorm.hip", "optim.hip", "attn_train.hip", "attn_flash.hip"]


STAMP = LIB + ".stamp"


def _sources_digest():
    """sha256 over the kernel sources, the ABI header and the compiler flags -- NOT modification times: the
    snapshot that carries the prebuilt library to the GPU box does not preserve them, and an unnecessary rebuild
    there would land inside the driver's timing of the first command that imports the package."""
    import hashlib
    h = hashlib.sha256()
    paths = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    paths.append(os.path.join(os.path.dirname(HERE), "include", "agile3d_hip.h"))
    paths.append(os.path.abspath(__file__))
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != _sources_digest()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  Without it hipcc runs the MFMAs in
        # AGPR form but keeps the loop-carried accumulators in VGPRs, copying all of them in and out
        # (v_accvgpr_write/read + hazard nops + an MFMA pipeline drain) every stage: 2x slower.
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"---- {src} failed ----\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_sources_digest() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
