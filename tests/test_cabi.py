"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/fastvim_hip.h declares, and the product has no CPU fallback."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fastvim_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import fastvim_amd.build as fb
    fb.build()
    from fastvim_amd import _lib
    lib = _lib.lib()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/fastvim_hip.h but not exported"
    assert sorted(_lib.C_ABI_SYMBOLS) == declared
    assert lib.fv_version() >= 1


def test_no_cpu_fallback():
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    u = torch.randn(1, 2, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        selective_scan_fn(u, u, -torch.rand(2, 4), torch.randn(1, 4, 4), torch.randn(1, 4, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "fastvim_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
