"""Build libstreammind_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m streammind_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libstreammind_hip.so")
SOURCES = ["linear.hip", "gemm256.hip", "wstream.hip", "gemm_fp8.hip", "attention.hip", "vecops.hip", "ingest.hip", "stc.hip", "comm.hip", "jpeg.hip", "model.hip"]
HEADERS = ["common.h", "host.h", "linear_common.h", os.path.join("..", "..", "include", "streammind_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _newer(src: str, dst: str) -> bool:
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    objdir = os.path.join(OUT_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    objs, rebuilt = [], False
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdr_paths):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
            rebuilt = True
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
