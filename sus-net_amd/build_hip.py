"""Compile sus-net_amd/csrc/susnet_capi.hip -> sus-net_amd/libsusnet_hip.so for gfx950 (in-tree)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "susnet_capi.hip")
DEPS = [SRC, os.path.join(PKG_DIR, "csrc", "susnet_device.h"), os.path.join(PKG_DIR, "csrc", "susnet_obs.h"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "susnet.h")]
OUT = os.path.join(PKG_DIR, "libsusnet_hip.so")


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-pass-failed", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
