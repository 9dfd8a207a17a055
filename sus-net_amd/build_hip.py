"""Compile sus-net_amd/csrc/*.hip -> sus-net_amd/libsusnet_hip.so for gfx950 (in-tree).

One object per translation unit (susnet_capi.hip = host side + the small kernels; inst_*.hip = one compiled-in
configuration of the stepping kernels each), compiled in parallel, then linked into ONE shared library."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "_obj")
OUT = os.path.join(PKG_DIR, "libsusnet_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-pass-failed"]
# -amdgpu-sched-strategy=max-ilp for the stepping kernels: they run at ONE wave per SIMD by design (a wave per 64 environments), so the
# scheduler's default goal -- registers for occupancy -- buys nothing; scheduled for instruction-level parallelism the same sources run
# faster (same-box A/B, gpurun_out/r05v, r05w, r05x: cfg3 +3-7 %, cfg4 +3-5 %, tag5 +1-2 %; the family within +- 2 %) and every ISA check
# below still passes.  Not for the Q-network units: their matrix sections are scheduled by hand (sched_barrier) and the network kernel
# alone came out 11 % slower under it (49.3 -> 55.1 us; the one-kernel tick unchanged).  Nor for the 1v1 unit: no-walls within noise
# (+0.7 %), the wall-map flavour 3 % slower (300.9 -> 292.1 G, gpurun_out/r05ad).  Nor for susnet_capi.hip: its kernels (feature rows,
# ring append, reset, observe) run many waves per SIMD and live on occupancy -- under max-ilp k_ring_append goes from 69 to 77 vector
# registers (7 -> 6 waves), k_featurize / k_observe from 83 to 105 (5 -> 4)
ILP_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def flags_for(src: str):
    name = os.path.basename(src)
    return FLAGS + (ILP_FLAGS if name.startswith("inst_") and not name.startswith(("inst_qnet", "inst_cfg2")) else [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(PKG_DIR), "include", "susnet.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    srcs, hdrs = sources(), headers()
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    todo, objs = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            todo.append((src, obj))
    for stale in set(glob.glob(os.path.join(OBJ_DIR, "*.o"))) - set(objs):
        os.remove(stale)
    if not todo and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(o) for o in objs):
        return OUT

    def compile_one(job):
        src, obj = job
        cmd = [cc, *flags_for(src), "-c", "-o", obj, src]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = jobs or int(os.environ.get("SUSNET_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_one, todo))
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    verify(OUT)
    return OUT


def verify(lib: str = OUT) -> None:
    """The ISA checks a linked library must pass before it ships (sus-net_amd/isa_checks.py): no wide-store data hazard (the
    gfx950 hazard BufDst::st128 works around: a toolchain update or a new call site must not bring it back silently), no
    scratch traffic, the register budget of the benchmarked kernels.  A failing library is removed."""
    if os.environ.get("SUSNET_SKIP_ISA_CHECKS") == "1":
        return
    try:
        from . import isa_checks
    except ImportError:  # run as a script
        import isa_checks
    if not isa_checks.tools_available():
        raise RuntimeError("llvm-objdump / llvm-readelf / c++filt not found: cannot verify the gfx950 code objects "
                           "(SUSNET_SKIP_ISA_CHECKS=1 builds without the check)")
    _, problems = isa_checks.verify_library(lib)
    if problems:
        os.replace(lib, lib + ".rejected")
        raise RuntimeError("libsusnet_hip.so failed the ISA checks (kept as " + lib + ".rejected):\n  " + "\n  ".join(problems))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
