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
LIB_DBG = os.path.join(HERE, "libsr_hotpath_dbg.so")     # diagnostic build (in-kernel stamps), tools/ only
SOURCES = ["sr_abi.hip"]
ARCH = "gfx950"


def _deps():
    out = [os.path.join(HERE, "..", "include", "sr_hotpath.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip")):
            out.append(os.path.join(CSRC, f))
    return out


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    """debug=True builds libsr_hotpath_dbg.so with -DSR_DEBUG_STAMPS (never loaded by the product path)."""
    LIB = LIB_DBG if debug else globals()["LIB"]
    if not force and not needs_build(LIB):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(HERE, "..", "include")]
    if debug:
        cmd += ["-DSR_DEBUG_STAMPS"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    report = parse_resource_usage(res.stderr)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        raise RuntimeError("hipcc failed")
    with open(os.path.join(HERE, "kernel_resources_dbg.txt" if debug else "kernel_resources.txt"), "w") as f:
        for k in report:
            f.write("{name} vgpr={vgpr} agpr={agpr} spill={spill} scratch={scratch} lds={lds} occ={occ}\n".format(**k))
    bad = [k for k in report if k["spill"] or k["scratch"]]
    if bad and debug:
        sys.stderr.write("diagnostic build: spills in " + ", ".join(k["name"] for k in bad) + " (timing only)\n")
    elif bad:
        # ROCm 7.2 hipcc miscompiles a partially spilled fragment (see csrc/sr_common.h): never ship a spill
        raise RuntimeError("register spills in: " + ", ".join(f"{k['name']} (spill {k['spill']})" for k in bad))
    os.replace(LIB + ".tmp", LIB)
    return LIB


def parse_resource_usage(text: str):
    import re
    out, cur = [], None
    pats = {"vgpr": r" VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "spill": r"VGPRs Spill: (\d+)",
            "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)",
            "occ": r"Occupancy \[waves/SIMD\]: (\d+)"}
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "vgpr": 0, "agpr": 0, "spill": 0, "scratch": 0, "lds": 0, "occ": 0}
            out.append(cur)
            continue
        if cur is None:
            continue
        for k, p in pats.items():
            m = re.search(p, line)
            if m:
                cur[k] = int(m.group(1))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
