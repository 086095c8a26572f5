"""Compile liblidarhip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is built into ``lidar_transfer_amd/lib/`` so that it travels with a
repository snapshot to the GPU box (a JIT cache under ~/.cache would not).
``-ffp-contract=off`` is part of the numerical contract: the triangle test must
round exactly like the reference's SSE code (no fused multiply-add).
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liblidarhip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Build the shared library if it is missing or stale; return its path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblidarhip.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-o", LIB_PATH, *sources()]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
