"""ctypes binding of libprv2_hip.so (the C ABI declared in include/prv2.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the
product raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C patchrefinerv2_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PRV2_HIP_LIB: probes / the negative-control test load a scratch build of the SAME C ABI -- e.g. coarse_taps.hip with packed fp32 math -- in a
#  child process; the product always loads the in-tree library)
LIB_PATH = os.environ.get("PRV2_HIP_LIB") or os.path.join(_HERE, "libprv2_hip.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_SOFTPLUS, ACT_SILU = 0, 1, 2, 3, 4, 5
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16F6 = 0, 1, 2, 3  # (F16F6: prv2_conv3x3_f6 only)
PREC_LABEL = {PREC_F32: "f32", PREC_BF16X3: "bf16x3", PREC_BF16: "bf16", PREC_F16F6: "f16f6"}
# model-level arithmetic names.  "f16f6" = bf16x3 everywhere, with the fp16 + block-scaled-fp6 product in the layers that have a kernel for it
# (stage 1: GatedConvUnit.conv, csrc/conv3x3_f6.hip) -- the modules map the name to PREC_BF16X3 and fusion.py looks at the name itself
PREC_NAMES = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "f16f6": PREC_BF16X3}
FMT_X_X2, FMT_MUL_X2, FMT_Y_X2 = 1, 2, 4  # prv2_conv_desc.fmt: operands in the pre-split "X2" activation format
ABI_VERSION = 19


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ldx", C.c_int32), ("ldy", C.c_int32),
        ("x_bstride", C.c_int64), ("y_bstride", C.c_int64),
        ("relu_in", C.c_int32), ("act", C.c_int32), ("convt_k", C.c_int32),
        ("ld_mul", C.c_int32), ("ld_res", C.c_int32), ("ld_res2", C.c_int32),
        ("prec", C.c_int32), ("force_generic", C.c_int32), ("ln_eps", C.c_float), ("part", C.c_int32),
        ("same_pad", C.c_int32), ("fmt", C.c_int32),
    ]


class UpsSrc(C.Structure):  # prv2_ups_src
    _fields_ = [("x", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32), ("ld", C.c_int32), ("channels", C.c_int32), ("bstride", C.c_int64)]


class Chain32Desc(C.Structure):  # prv2_chain32_desc
    _fields_ = [("x", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("wg", C.c_void_p), ("wo", C.c_void_p), ("consts", C.c_void_p),
                ("pre", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("y", C.c_void_p), ("depth", C.c_void_p),
                ("x_bstride", C.c_int64), ("y_bstride", C.c_int64), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("ldx", C.c_int32), ("ldy", C.c_int32), ("ld_pre", C.c_int32), ("b3", C.c_float), ("ln_eps", C.c_float)]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/prv2.h declares
SIGNATURES = {
    "prv2_abi_version": (_I, []),
    "prv2_last_error": (C.c_char_p, []),
    "prv2_last_kernel": (C.c_char_p, []),
    "prv2_packed_weight_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "prv2_pack_conv_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "prv2_conv2d": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prv2_conv2d_ups_supported": (_I, [C.POINTER(ConvDesc), C.POINTER(UpsSrc)]),
    "prv2_conv2d_ups": (_I, [C.POINTER(ConvDesc), _P, C.POINTER(UpsSrc), _P, _P, _P, _P, _P, _P, _P]),
    "prv2_conv2d_tail_supported": (_I, [C.POINTER(ConvDesc)]),
    "prv2_conv2d_tail": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "prv2_conv2d_cout1": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _F, _P, _I, _P, _P]),
    "prv2_dwconv2d": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _P]),
    "prv2_dwconv2d_ex": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "prv2_global_avgpool_workspace_floats": (_L, [_I, _L, _I]),
    "prv2_global_avgpool": (_I, [_P, _I, _L, _I, _I, _P, _P, _P]),
    "prv2_se_gate": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    "prv2_channel_scale": (_I, [_P, _I, _L, _I, _I, _P, _P]),
    "prv2_layernorm": (_I, [_P, _L, _I, _I, _P, _P, _F, _I, _P, _I, _P]),
    "prv2_patchify": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "prv2_assemble_tokens": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "prv2_attention": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _L, _P]),
    "prv2_attention_bias_image_bytes": (_L, [_I, _I]),
    "prv2_pack_attention_bias": (_I, [_P, _I, _I, _I, _P, _P]),
    "prv2_attention_bias": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _L, _P]),
    "prv2_split_ss": (_I, [_P, _L, _I, _I, _P, _P]),
    "prv2_layernorm_ss": (_I, [_P, _L, _I, _I, _P, _P, _F, _P, _P]),
    "prv2_attention_ss": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _L, _P]),
    "prv2_gemm_ss": (_I, [_P, _L, _I, _P, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P]),
    "prv2_gemm_ss_qkv": (_I, [_P, _L, _I, _P, _I, _P, _I, _F, _P, _P]),
    "prv2_attention_qkv_ss": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "prv2_attention_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "prv2_bicubic_resize": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "prv2_add": (_I, [_P, _I, _P, _I, _L, _I, _P, _I, _P]),
    "prv2_zero_pad_channels": (_I, [_P, _L, _I, _I, _P]),
    "prv2_zoe_attractor": (_I, [_P, _I, _I, _P, _I, _I, _F, _L, _P, _I, _P]),
    "prv2_zoe_logbinom_depth": (_I, [_P, _I, _P, _I, _I, _F, _F, _L, _P, _P]),
    "prv2_crop_resize": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _I, _P]),
    "prv2_roi_align": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _I, _I, _P, _I, _P]),
    "prv2_roi_align_x2": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _I, _I, _P, _I, _P]),
    "prv2_upsample_bilinear": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "prv2_conv3x3_ln_gate_supported": (_I, [_P]),
    "prv2_gate_weight_bytes": (C.c_int64, [_I]),
    "prv2_pack_gate_weight": (_I, [_P, _P, _I, _I, _P]),
    "prv2_conv3x3_ln_gate": (_I, [_P] * 12),
    "prv2_conv3x3_ln_gate_pre": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prv2_upconv3x3_supported": (_I, [C.POINTER(UpsSrc), _I, _I, _I, _I, _I]),
    "prv2_upconv3x3": (_I, [C.POINTER(UpsSrc), _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _L, _P]),
    "prv2_upconv5x5_supported": (_I, [C.POINTER(UpsSrc), _I, _I, _I, _I, _I]),
    "prv2_upconv5x5": (_I, [C.POINTER(UpsSrc), _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _L, _P]),
    "prv2_upconv5x5_lines": (_I, [C.POINTER(UpsSrc), _I, _I, _I, _P, _P]),
    "prv2_upconv5x5_ring": (_I, [_P, _I, _L, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P]),
    "prv2_conv2d_pre_supported": (_I, [C.POINTER(ConvDesc)]),
    "prv2_conv2d_pre": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "prv2_chain32_weight_bytes": (_L, [_I, _I]),
    "prv2_pack_chain32_weight": (_I, [_P, _I, _I, _I, _P, _P]),
    "prv2_chain32_c2f": (_I, [C.POINTER(Chain32Desc), _P]),
    "prv2_chain32_enc": (_I, [C.POINTER(Chain32Desc), _P]),
    "prv2_conv3x3_f6_supported": (_I, [C.POINTER(ConvDesc)]),
    "prv2_conv3x3_f6_weight_bytes": (_L, [_I, _I]),
    "prv2_pack_conv3x3_f6_weight": (_I, [_P, _F, _P, _I, _I, _P]),
    "prv2_conv3x3_f6": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _F, _F, _P, _P, _P]),
    "prv2_conv3x3_ln_gate_f6": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P]),
    "prv2_coarse_tap_knots": (_I, [_P, _I, _I, _I, _I, _F, _F, _P, _I, _P]),
    "prv2_coarse_tap_gather": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _I, _F, _I, _I, _P, _I, _P]),
    "prv2_conv_border_bias": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "prv2_depth_pair_fill": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "prv2_nchw_to_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "prv2_nhwc_to_nchw": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "prv2_blend_paste": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "prv2_blend_update": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "prv2_blend_resize": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or fail loudly (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7 / libhsa-runtime64; it must be the first HIP runtime in the
    # process (same SONAME as /opt/rocm's), otherwise the two halves of the runtime mix and launches fail.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP library has not been built "
            "(run `make -C patchrefinerv2_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.prv2_abi_version() != ABI_VERSION:
        raise ImportError(f"libprv2_hip.so ABI {lib.prv2_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().prv2_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")
