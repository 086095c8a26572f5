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
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 operations into v_pk_* instructions and pays for them with as many
# v_mov to line the operands up -- k_sc_tris is 11 % faster without it.
# LIDARHIP_EXTRA_FLAGS (space separated) is appended: compiler experiments on the GPU box.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-result",
         "-fno-slp-vectorize", *os.environ.get("LIDARHIP_EXTRA_FLAGS", "").split()]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


HASH_PATH = LIB_PATH + ".srchash"


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


def source_hash() -> str:
    """Content hash of everything the library is compiled from (and of the flags)."""
    import hashlib
    h = hashlib.sha256(" ".join([ARCH, *FLAGS]).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    """Stale = built from other sources.  Decided by content, not by file times: a repository snapshot copied to
    another machine keeps its bytes but not necessarily the order of its mtimes, and a spurious rebuild there
    would have every rank of a multi-GPU job compile the same file at once."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(HASH_PATH) as f:
            return f.read().strip() != source_hash()
    except OSError:
        # no record (library built by hand): fall back to file times
        t = os.path.getmtime(LIB_PATH)
        return any(os.path.getmtime(d) > t for d in _deps())


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Build the shared library if it is missing or stale; return its path.  Safe against concurrent callers
    (ranks of one job): one compiles under a file lock into a temporary name and renames, the others wait."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblidarhip.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # somebody else built it while we waited
                return LIB_PATH
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                   "-o", tmp, *sources()]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIB_PATH)
            with open(HASH_PATH, "w") as f:
                f.write(source_hash() + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
