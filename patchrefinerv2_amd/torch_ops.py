"""``torch.ops.prv2.*``: the hot path's operators as PyTorch-ROCm custom ops (TORCH_LIBRARY(prv2), csrc/torch_ops.cpp).

This is the operator surface BASELINE.json's north star names ("surfaced to Python as PyTorch-ROCm custom ops"): every op
takes / returns ``at::Tensor``, runs on torch's current HIP stream, allocates through the caching allocator, raises
RuntimeError through TORCH_CHECK, and sits directly on the C ABI of include/prv2.h (libprv2_hip.so).  Registered for the
CUDA (= HIP) dispatch key only: a CPU tensor is rejected by the dispatcher -- there is no CPU implementation to fall into.

    from patchrefinerv2_amd import torch_ops; torch_ops.load()
    y = torch.ops.prv2.conv2d(x_nhwc, w_packed, bias, cout, 3, 3, pad=1, act=torch_ops.ACT_GELU, prec=torch_ops.PREC_BF16X3)

Activations are NHWC float32; a channel slice of a wider NHWC buffer (``buf[..., c0:c0 + c]``) is a valid input and a valid
``out=``, which is how the reference's ``torch.cat`` calls are written in place.
"""
from __future__ import annotations

import os

from .lib import (ABI_VERSION, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_SOFTPLUS, PREC_BF16, PREC_BF16X3,  # noqa: F401
                  PREC_F32)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprv2_torch.so")
OPS = ("abi_version", "pack_conv_weight", "conv2d", "conv3x3_ups", "pack_gate_weight", "conv3x3_ln_gate", "layernorm", "attention_fwd", "crop_resize_bilinear", "roi_gather_pyramid", "roi_align",
       "upsample_bilinear_ac", "blend_init", "blend_update", "blend_resize", "zoe_attractor", "zoe_bins_head", "nchw_to_nhwc",
       "nhwc_to_nchw",
       # round 4: every remaining entry point a frame uses (the host mirror's default route can be torch.ops: ops.DISPATCH)
       "conv3x3_tail", "conv3x3_pre", "coarse_tap_knots", "coarse_tap_gather", "conv_cout1", "dwconv2d", "global_avgpool", "se_gate", "channel_scale_",
       "patchify", "assemble_tokens", "split_ss", "layernorm_ss", "gemm_ss", "attention_ss", "bicubic_resize", "depth_pair_fill", "conv_border_bias_",
       "add_nhwc", "zero_pad_channels_", "pack_attention_bias", "upconv3x3",
       # round 5: the fused 32-channel full-resolution chains and the 5x5 composite at the source resolution
       "pack_chain32_weight", "chain32_c2f", "chain32_enc", "upconv5x5", "upconv5x5_lines", "upconv5x5_ring_", "pack_conv3x3_f6_weight", "conv3x3_f6", "conv3x3_ln_gate_f6")
_loaded = False


def load():
    """Register the ops (idempotent) or fail loudly: ImportError when the shim has not been built."""
    global _loaded
    import torch
    if _loaded:
        return torch.ops.prv2
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C patchrefinerv2_amd/csrc` "
                          "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback.")
    torch.ops.load_library(LIB_PATH)
    if torch.ops.prv2.abi_version() != ABI_VERSION:
        raise ImportError(f"libprv2_torch.so sits on C ABI {torch.ops.prv2.abi_version()}, expected {ABI_VERSION}; rebuild")
    _loaded = True
    return torch.ops.prv2
