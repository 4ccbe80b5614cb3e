#!/usr/bin/env python3
"""Builds libnnue_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The library is placed in-tree next to the host layer (nnue-vision_amd/nnue_hip/) so that it
travels with the source snapshot; it is git-ignored.  Rebuilds only when a source is newer.
"""
import os
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
ROOT = CSRC.parent.parent
OUT = CSRC.parent / "nnue_hip" / "libnnue_hip.so"
SOURCES = ["abi.cpp", "ft_kernels.hip", "ftb_kernels.hip", "ftm_kernels.hip", "feature_kernels.hip", "classifier_kernels.hip", "optim_kernels.hip", "input_kernels.hip", "engine_kernels.hip", "exchange_kernels.hip", "ftv_kernels.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
         "-fno-gpu-rdc", "-x", "hip", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
# NNUE_BUILD_ABLATIONS=1 also compiles the timing-only ablation variants tools/debug uses (NNUE_*_ABL* knobs: wrong results by
# design, so a default build does not contain them)
if os.environ.get("NNUE_BUILD_ABLATIONS") == "1":
    FLAGS.append("-DNNUE_ABLATIONS")
OBJ = CSRC / "build"  # per-source objects (git-ignored): only changed sources are recompiled, in parallel


def stale() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + sorted(CSRC.glob("*.h")) + [ROOT / "include" / "nnue_hip.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not stale():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OBJ.mkdir(exist_ok=True)
    shared = sorted(CSRC.glob("*.h")) + [ROOT / "include" / "nnue_hip.h", Path(__file__)]  # any header rebuilds every object
    newest_shared = max(d.stat().st_mtime for d in shared)

    def compile_one(src: str) -> Path:
        obj = OBJ / (src.rsplit(".", 1)[0] + ".o")
        if force or not obj.exists() or obj.stat().st_mtime < max((CSRC / src).stat().st_mtime, newest_shared):
            cmd = [hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
