"""The C-ABI library loads without a GPU and exports every symbol include/sr_hotpath.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mobilesuperresolution_amd import _lib, build
    lib = build.build()
    h = ctypes.CDLL(lib)
    header = open(os.path.join(ROOT, "include", "sr_hotpath.h")).read()
    declared = set(re.findall(r"\b(?:int|void|sr_graph_cache_t\*)\s+(sr_\w+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(h, name), f"{name} declared in sr_hotpath.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert h.sr_abi_version() == _lib.ABI_VERSION


def test_no_spills_in_any_kernel():
    from mobilesuperresolution_amd import build
    build.build()
    txt = open(os.path.join(ROOT, "mobilesuperresolution_amd", "kernel_resources.txt")).read()
    assert "spill=0" in txt and not re.search(r"spill=[1-9]", txt) and not re.search(r"scratch=[1-9]", txt)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mobilesuperresolution_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dp, f)).read(), f
