"""CPU tests of the drop-in boundary: the C-ABI library builds, loads, exports every symbol include/sylph_hip.h
declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

import sylph_amd
from sylph_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sylph_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sylph_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = ctypes.CDLL(binding.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sylph_hip.h but not exported"
    assert sorted(binding.EXPORTS) == syms


def test_version_and_error_string():
    L = sylph_amd.load()
    assert L.sylph_version() >= 100
    assert isinstance(L.sylph_last_error(), bytes)


def test_product_never_touches_the_oracle():
    """The product path (sylph_amd/, include/) must not import, link or call anything under oracle/."""
    bad = []
    for d in ("sylph_amd", "include"):
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                    txt = open(os.path.join(base, f), errors="replace").read()
                    if re.search(r"oracle[/.]|liboracle|from oracle|import oracle", txt):
                        bad.append(os.path.join(base, f))
    assert not bad, bad


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sylph_amd.SylphHipError):
        sylph_amd.Context(0)


def test_reference_side_binding_covers_every_entry_point():
    """INTEGRATION.md and integration/rust/hip_ffi.rs (the binding a sylph maintainer would add) declare every function of
    include/sylph_hip.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sylph_hip.h")).read()
    names = sorted(set(re.findall(r"\b(sylph_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 30
    for doc in ("INTEGRATION.md", os.path.join("integration", "rust", "hip_ffi.rs")):
        text = open(os.path.join(root, doc)).read()
        missing = [n for n in names if n not in text]
        assert not missing, (doc, missing)
