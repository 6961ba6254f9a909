"""In-tree build of the HIP library: hipcc cross-compiles gfx950 without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "spf_hip.hip")
OUT_DIR = os.path.join(_HERE, "lib")
OUT = os.path.join(OUT_DIR, "libspf_hip.so")

# -ffp-contract=off is part of the numerical contract (see csrc/spf_device.hpp)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC",
               "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _sources():
    d = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include", "spf_hip.h")
    return [p for p in (os.path.join(d, f) for f in os.listdir(d)) if os.path.isfile(p)] + [inc]


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in _sources())


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile spf_amd/csrc/spf_hip.hip -> spf_amd/lib/libspf_hip.so (gfx950)."""
    if not force and not is_stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    # compile beside the target and rename atomically, so a concurrent loader (another rank of a
    # multi-GPU launch) never sees a half-written library
    tmp = f"{OUT}.{os.getpid()}.tmp"
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT
