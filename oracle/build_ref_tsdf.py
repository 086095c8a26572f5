#!/usr/bin/env python3
"""oracle/build_ref_tsdf.py -- TEST INFRASTRUCTURE: builds ``oracle/_ref/libref_tsdf_integrate.so`` from the reference's
OWN TSDF kernel.

The reference's class-aware ``integrate`` is CUDA C held in a Python string
(/root/reference/auxiliary/fusion_lidar.py:66-229, the argument of pycuda's ``SourceModule``).  This recipe reads that
string where it lies (``ast`` on the file: nothing of the reference is imported or executed, pycuda is not needed), puts it
into a scratch file OUTSIDE the repository with pycuda's default ``extern "C"`` wrapping and our launch code
(oracle/ref_tsdf_launch.inc) behind it, and lets hipcc compile it for gfx950.  Only the .so lands in ``oracle/_ref/``
(git-ignored; travels to the GPU box like libref_strict.so); the scratch file is deleted.

Compiler flags = nvcc's defaults restated for hipcc: ``-ffp-contract=fast`` (nvcc -fmad=true), IEEE division and square
root (hipcc's default), no flush-to-zero.  What this does NOT give: CUDA's math library.  norm3df / atan2 / asinf resolve to
HIP's device library, so this pins the restatements (oracle/lt_tsdf_dense.hip, lt_tsdf_oracle.c, the product kernels) to
the reference's SOURCE TEXT -- every branch, cast, rounding call and update rule -- not to NVIDIA's last-ulp behaviour of
three math functions.
"""
import ast
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PY = os.environ.get("LT_REF_FUSION", "/root/reference/auxiliary/fusion_lidar.py")
OUT = os.path.join(HERE, "_ref", "libref_tsdf_integrate.so")
# the same text with the reference's own hard-wired switch (`bool merge = true;`, fusion_lidar.py:157) flipped -- the one
# token that selects the plain running average (:158-...) instead of the class-aware update; nothing else is touched
OUT_PLAIN = os.path.join(HERE, "_ref", "libref_tsdf_integrate_plain.so")
SWITCH = "bool merge = true;"


def kernel_text(path=REF_PY):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", None)) == "SourceModule":
            return ast.literal_eval(node.args[0]), node.args[0].lineno, node.args[0].end_lineno
    raise SystemExit(f"{path}: no SourceModule(...) call found")


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(REF_PY):
        print("oracle: reference absent, keeping prebuilt", OUT)
        return 0
    if not os.path.exists(hipcc):
        print("oracle: no hipcc, keeping prebuilt", OUT)
        return 0
    text, a, b = kernel_text()
    if text.count(SWITCH) != 1:
        raise SystemExit(f"{REF_PY}: expected exactly one `{SWITCH}` in the kernel text")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    for out, body, note in ((OUT, text, ""), (OUT_PLAIN, text.replace(SWITCH, "bool merge = false;"), " (merge switch flipped)")):
        with tempfile.TemporaryDirectory(prefix="lt_ref_tsdf_") as tmp:
            src = os.path.join(tmp, "ref_integrate.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n")
                f.write(f"// {REF_PY}:{a}-{b}{note}, wrapped as pycuda's SourceModule does (no_extern_c=False)\n")
                f.write('extern "C" {\n' + body + "\n}\n")
                f.write(f'#include "{os.path.join(HERE, "ref_tsdf_launch.inc")}"\n')
            cmd = [hipcc, "-O3", "-std=c++17", "-ffp-contract=fast", "-fPIC", "-shared", "-w", "--offload-arch=gfx950",
                   "-o", out + ".tmp", src]
            subprocess.run(cmd, check=True)
            os.replace(out + ".tmp", out)
        print("oracle: built", out, f"from {REF_PY}:{a}-{b}{note}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
