"""Build the TEST-ONLY native helpers under tests/csrc/ into tests/lib/ (in-tree, so that they travel to the GPU box
with the snapshot; hipcc cross-compiles gfx950 without a GPU).  Nothing in lidar_transfer_amd/ links or loads these.

    lt_tsdf_dense.hip -> liblt_tsdf_dense.so   one-thread-per-voxel restatement of the reference's pycuda `integrate`
                                              kernel: A/B partner of the product's column-aware kernel
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "--offload-arch=gfx950"]
HELPERS = {"liblt_tsdf_dense.so": ["lt_tsdf_dense.hip"]}


def _hash(srcs):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for s in srcs:
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + f.read())
    return h.hexdigest()


def build(name: str, force: bool = False) -> str:
    srcs = HELPERS[name]
    out = os.path.join(LIB, name)
    stamp = out + ".srchash"
    want = _hash(srcs)
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError(f"hipcc not found: cannot build {name}")
    os.makedirs(LIB, exist_ok=True)
    tmp = out + f".tmp{os.getpid()}"
    res = subprocess.run([hipcc, *FLAGS, "-o", tmp, *[os.path.join(CSRC, s) for s in srcs]], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{res.stdout}{res.stderr}")
    os.replace(tmp, out)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return out


def build_all(force: bool = False):
    return [build(n, force) for n in HELPERS]


if __name__ == "__main__":
    print(build_all(force=True))
