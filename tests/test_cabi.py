"""CPU: the C-ABI shared library loads and exports every symbol include/prv2.h declares
(no compute calls without a GPU); argument validation fails loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "prv2.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(prv2_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_header_symbols():
    from patchrefinerv2_amd import lib as L
    names = _declared()
    assert len(names) >= 30
    assert set(names) == set(L.SIGNATURES), set(names) ^ set(L.SIGNATURES)
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert L.load().prv2_abi_version() == L.ABI_VERSION


def test_conv_desc_layout_matches_header():
    from patchrefinerv2_amd import lib as L
    txt = open(os.path.join(ROOT, "include", "prv2.h")).read()
    body = txt[txt.index("typedef struct prv2_conv_desc {"):txt.index("} prv2_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"(?:int(?:32|64)_t|float)\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in L.ConvDesc._fields_]
    assert ctypes.sizeof(L.ConvDesc) == 112


def test_rejects_bad_arguments_without_gpu():
    from patchrefinerv2_amd import lib as L
    lib = L.load()
    assert lib.prv2_layernorm(None, 4, 8, 8, None, None, 1e-6, 0, None, 8, None) != 0
    assert b"null" in lib.prv2_last_error()
    with pytest.raises(RuntimeError):
        L.check(lib.prv2_attention(None, 1, 1, 1, 64, None, 0, None, 0, None), "attention")
    assert lib.prv2_packed_weight_bytes(256, 514, 3, 3, 0, 0) == 256 * 9 * 544 * 4


def test_product_does_not_import_oracle():
    """the shipped package must never route through the oracle"""
    pkg = os.path.join(ROOT, "patchrefinerv2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
