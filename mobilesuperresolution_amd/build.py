"""Build libsr_hotpath.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m mobilesuperresolution_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsr_hotpath.so")
SOURCES = ["sr_abi.hip"]
ARCH = "gfx950"


def _deps():
    out = [os.path.join(HERE, "..", "include", "sr_hotpath.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip")):
            out.append(os.path.join(CSRC, f))
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", os.path.join(HERE, "..", "include")]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
