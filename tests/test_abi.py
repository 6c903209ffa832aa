"""CPU-only checks of the C ABI: the shared library loads, exports every symbol that
include/triplets_amd.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import triplets_amd as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "triplets_amd.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trp_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(T.LIB_PATH):
        pytest.skip("library not built here (run __graft_entry__.build())")
    L = T.load_library()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert set(T.EXPORTS) == set(names)


def test_no_cpu_fallback():
    if not os.path.exists(T.LIB_PATH):
        pytest.skip("library not built here")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(T.TripletsAmdError):
        T.Context(0)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure; nothing under the package may import or link it."""
    pkg = os.path.join(ROOT, "rust-triplets_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    txt = fh.read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "orc_" not in txt, f
