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


def test_chain32_desc_layout_matches_header():
    """prv2_chain32_desc (round 5): field order / sizes of the ctypes mirror == the header's struct"""
    from patchrefinerv2_amd import lib as L
    txt = open(os.path.join(ROOT, "include", "prv2.h")).read()
    body = txt[txt.index("typedef struct prv2_chain32_desc {"):txt.index("} prv2_chain32_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"(?:const\s+)?(?:float|void|int(?:32|64)_t)\s*\*?\s*([^;]+);", body):
        fields += [f.strip().lstrip("*") for f in decl.split(",")]
    assert fields == [f[0] for f in L.Chain32Desc._fields_], fields
    assert ctypes.sizeof(L.Chain32Desc) == 11 * 8 + 2 * 8 + 6 * 4 + 2 * 4 == 136


def test_round5_entry_points_reject_bad_arguments_without_gpu():
    from patchrefinerv2_amd import lib as L
    lib = L.load()
    assert lib.prv2_chain32_c2f(None, None) != 0 and b"null" in lib.prv2_last_error()
    d = L.Chain32Desc(x=16, w1=16, w2=16, wg=0, wo=0, consts=16, y=16, n=1, h=8, w=16, ldx=30, ldy=32)
    assert lib.prv2_chain32_c2f(ctypes.byref(d), None) != 0 and b"32-channel" in lib.prv2_last_error()       # ldx < 32
    d.ldx = 32
    assert lib.prv2_chain32_c2f(ctypes.byref(d), None) != 0 and b"gate / out_conv" in lib.prv2_last_error()  # fragment images missing
    assert lib.prv2_chain32_enc(ctypes.byref(d), None) != 0 and b"tail fragment" in lib.prv2_last_error()
    assert lib.prv2_chain32_weight_bytes(0, 9) == 9 * 2 * 2 * 64 * 16 and lib.prv2_chain32_weight_bytes(2, 9) == 2 * 2 * 64 * 16
    us = L.UpsSrc(x=16, h=12, w=16, ld=256, channels=256, bstride=0)
    assert lib.prv2_upconv5x5_supported(ctypes.byref(us), 1, 24, 32, 32, L.PREC_BF16X3) == 1
    assert lib.prv2_upconv5x5_supported(ctypes.byref(us), 1, 24, 32, 64, L.PREC_BF16X3) == 0      # cout > 32
    assert lib.prv2_upconv5x5_supported(ctypes.byref(us), 1, 20, 32, 32, L.PREC_BF16X3) == 0      # source step 11 / 19 > 1/2
    assert lib.prv2_upconv5x5_supported(ctypes.byref(us), 1, 24, 32, 32, L.PREC_F32) == 0
    assert lib.prv2_upconv5x5(ctypes.byref(us), None, None, 1, 24, 32, 32, 0, L.PREC_BF16X3, None, 32, 0, None) != 0
    assert lib.prv2_upconv5x5_ring(None, 32, 0, 1, 24, 32, 32, None, 896, 12, 16, 0, None) != 0
    # prv2_conv3x3_f6* (the fp16 + fp6 conv): shape contract, weight image size, scale / format validation
    mk = lambda **k: L.ConvDesc(**{**dict(n=1, h=24, w=32, cin=256, cout=256, kh=3, kw=3, stride=1, pad=1, ldx=256, ldy=256, x_bstride=0, y_bstride=0, relu_in=1,  # noqa: E731
                                        act=0, convt_k=0, ld_mul=0, ld_res=256, ld_res2=0, prec=L.PREC_F16F6, force_generic=0, ln_eps=1e-6, part=0, same_pad=0, fmt=0), **k})
    assert lib.prv2_conv3x3_f6_supported(ctypes.byref(mk())) == 1
    assert lib.prv2_conv3x3_f6_supported(ctypes.byref(mk(cin=96, ldx=96))) == 0 and lib.prv2_conv3x3_f6_supported(ctypes.byref(mk(cout=128))) == 0
    assert lib.prv2_conv3x3_f6_supported(ctypes.byref(mk(w=12))) == 0 and lib.prv2_conv3x3_f6_supported(ctypes.byref(mk(stride=2))) == 0
    assert lib.prv2_conv3x3_f6_weight_bytes(256, 256) == 4 * 9 * 65536 and lib.prv2_conv3x3_f6_weight_bytes(256, 96) == 0
    assert lib.prv2_pack_conv3x3_f6_weight(16, 3.0, 16, 256, 256, None) != 0 and b"power of two" in lib.prv2_last_error()
    assert lib.prv2_conv3x3_f6(ctypes.byref(mk()), 16, 16, None, None, 0.75, 1.0, None, 16, None) != 0 and b"power of two" in lib.prv2_last_error()
    assert lib.prv2_conv3x3_f6(ctypes.byref(mk(act=1)), 16, 16, None, None, 1.0, 1.0, None, 16, None) != 0 and b"no activation" in lib.prv2_last_error()
    assert lib.prv2_conv3x3_f6(ctypes.byref(mk(fmt=1)), 16, 16, None, None, 1.0, 1.0, None, 16, None) != 0   # only PRV2_FMT_Y_X2
    assert lib.prv2_conv3x3_f6(ctypes.byref(mk()), 16, 16, None, 8, 1.0, 1.0, None, 16, None) != 0 and b"res layout" in lib.prv2_last_error()


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
