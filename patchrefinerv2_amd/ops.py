"""Host-side wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

``Feat`` is an NHWC activation view (optionally a channel slice of a wider buffer) -- the
mechanism that makes every ``torch.cat`` of the reference free: producers write into slices.
"""
from __future__ import annotations

import ctypes as C
import math
import numpy as np
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import lib as L
from .lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_SOFTPLUS, PREC_F32  # noqa: F401


def _stream():
    return torch.cuda.current_stream().cuda_stream


# How the host mirror reaches the kernels: "torch" = through the PyTorch-ROCm custom ops torch.ops.prv2.* (csrc/torch_ops.cpp ->
# libprv2_torch.so: at::Tensor in / out on torch's current stream), "ctypes" = straight through the C ABI (lib.py).  Every entry
# point a frame uses exists on both routes; same kernels, same results bit for bit (tests/test_hip_models.py::
# test_models_through_torch_custom_ops).  PRV2_DISPATCH selects the default (INTEGRATION.md has the measured per-frame difference).
DISPATCH = os.environ.get("PRV2_DISPATCH", "torch")
assert DISPATCH in ("ctypes", "torch"), DISPATCH


def _tops():
    from . import torch_ops
    return torch_ops.load()


class Profiler:
    """Optional per-launch accounting for bench.py: algorithmic FLOPs (2*MAC) and, when ``timed``,
    HIP events recorded on the launch stream around each matrix-kernel launch."""

    def __init__(self):
        self.enabled = False
        self.timed = False
        self.by_shape = False
        self.records = []  # (kernel tag, executed flops, start event, end event, reference-graph flops)

    def start(self, timed=False, by_shape=False):
        self.enabled, self.timed, self.by_shape, self.records = True, timed, by_shape, []

    def stop(self):
        self.enabled = False
        return self.records

    def launch(self, tag, flops, fn, shape=None, algo=None):
        """``tag``: a string, or a callable evaluated AFTER the launch (the library reports which kernel it dispatched).
        ``flops``: the 2*MAC the launch EXECUTES; ``algo``: the 2*MAC the reference graph spends on the same step when that differs
        (the coarse half of a cat([fine, coarse_roi]) conv is computed once per frame here: coarse_tap_*) -- default: the same."""
        if not self.enabled:
            return fn()
        e0 = e1 = None
        if self.timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
        else:
            fn()
        if callable(tag):
            tag = tag()
        if self.by_shape and shape is not None:
            tag = f"{tag} {shape}"
        self.records.append((tag, flops, e0, e1, flops if algo is None else algo))

    def launch_aux(self, tag, nbytes, fn, shape):
        """memory-bound kernels: only itemised in the per-layer report (by_shape); bytes go into the tag"""
        if not (self.enabled and self.by_shape):
            return fn()
        return self.launch(tag, 0.0, fn, f"{shape} {nbytes / 1e6:.0f}MB")

    def summary(self):
        """{tag: dict(launches, flops (executed), algo (reference graph), ms)} (call after torch.cuda.synchronize())."""
        out = {}
        for tag, fl, e0, e1, al in self.records:
            d = out.setdefault(tag, dict(launches=0, flops=0.0, ms=0.0, algo=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["algo"] += al
            if e0 is not None:
                d["ms"] += e0.elapsed_time(e1)
        return out


PROFILER = Profiler()


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, Feat):
        return t.ptr
    return t.data_ptr()


def _require_dev(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise ValueError("prv2 ops need float32 tensors on the GPU (no CPU fallback exists)")
        if t is not None and t.device.index != torch.cuda.current_device():
            # kernels are enqueued on the CURRENT device's stream: a tensor of another device would be touched from the
            # wrong queue (the frame drivers enter torch.cuda.device(model.device) themselves)
            raise RuntimeError(f"prv2 op on a {t.device} tensor while the current device is cuda:{torch.cuda.current_device()}: "
                               "wrap the call in torch.cuda.device(tensor.device)")


def roundup(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _feat_of(y: torch.Tensor, c: int) -> "Feat":
    """the ``Feat`` of an NHWC tensor a torch.ops.prv2 operator allocated: dense, or -- channel counts that are not a multiple of 4 -- the
    first c channels of a buffer whose pixel stride is padded (csrc/torch_ops.cpp::alloc_nhwc)"""
    n, h, w, _ = y.shape
    ld = y.stride(2)
    return Feat(y if ld == y.shape[3] else torch.as_strided(y, (n, h, w, ld), (h * w * ld, w * ld, ld, 1)), c)


class Feat:
    """NHWC fp32 activation [n, h, w, c] living in ``buf`` ([n, h, w, ld]) at channel offset c0."""

    __slots__ = ("buf", "n", "h", "w", "c", "c0", "x2", "aux")

    def __init__(self, buf: torch.Tensor, c: Optional[int] = None, c0: int = 0, x2: bool = False):
        assert buf.dim() == 4 and buf.is_contiguous()
        _require_dev(buf)
        self.buf = buf
        # x2: the bytes are the pre-split "X2" operand format of include/prv2.h (per 8 channels [8 bf16 hi | 8 bf16 lo]) instead of
        # fp32 -- written by a producer for the 256-column conv kernels, its only readers (csrc/conv3x3_gate.hip)
        self.x2 = x2
        self.aux = None  # per-frame derived tensors a consumer attached to this map (fusion.BiDirectionalFusion.prepare_frame)
        self.n, self.h, self.w = buf.shape[0], buf.shape[1], buf.shape[2]
        self.c = buf.shape[3] - c0 if c is None else c
        self.c0 = c0
        assert 0 <= c0 and c0 + self.c <= buf.shape[3]

    @staticmethod
    def alloc_raw(n, h, w, c, device, pad_to: int = 4) -> "Feat":
        """like ``alloc`` but the pad channels are left to the caller (a producer that writes them: ops.depth_pair_fill)"""
        return Feat(torch.empty((n, h, w, roundup(c, pad_to)), device=device, dtype=torch.float32), c)

    @staticmethod
    def alloc(n, h, w, c, device, pad_to: int = 4) -> "Feat":
        ld = roundup(c, pad_to)
        # pad channels are read by the conv loader (x zero weights): they must be finite -- zero just those
        buf = torch.empty((n, h, w, ld), device=device, dtype=torch.float32)
        if ld != c:
            if DISPATCH == "torch":
                _tops().zero_pad_channels_(buf, c)
            else:
                L.check(L.load().prv2_zero_pad_channels(buf.data_ptr(), n * h * w, c, ld, _stream()), "zero_pad_channels")
        return Feat(buf, c)

    @property
    def ld(self) -> int:
        return self.buf.shape[3]

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + 4 * self.c0

    @property
    def device(self):
        return self.buf.device

    def slice(self, c0: int, c: int) -> "Feat":
        assert not self.x2 or (c0 % 8 == 0 and c % 8 == 0)
        return Feat(self.buf, c, self.c0 + c0, self.x2)

    def view(self) -> torch.Tensor:
        """this activation as a (strided) torch tensor [n, h, w, c]: what the torch.ops.prv2 operators take"""
        assert not self.x2, "a pre-split (X2) buffer is not an fp32 tensor"
        return self.buf[..., self.c0:self.c0 + self.c]

    def raw(self) -> torch.Tensor:
        """the same strided tensor whatever the format (an X2 buffer is a float32 CONTAINER: the operator is told by its ``fmt``)"""
        return self.buf[..., self.c0:self.c0 + self.c]

    def batch(self, b0: int, b1: int) -> "Feat":
        return Feat(self.buf[b0:b1], self.c, self.c0, self.x2)

    def x2_to_float(self) -> "Feat":
        """tests / debugging: the fp32 values an X2 buffer stands for (hi + lo per element)"""
        assert self.x2 and self.c0 % 8 == 0 and self.c % 8 == 0
        raw = self.buf[..., self.c0:self.c0 + self.c].contiguous().view(torch.int32)              # [n, h, w, c] dwords
        g = raw.view(self.n, self.h, self.w, self.c // 8, 2, 4)                                   # [.., group, hi | lo, 4 dwords = 8 bf16]
        def unpack(d):                                                                            # 4 dwords -> 8 floats (element 2i = low half)
            lo16 = (d << 16).view(torch.float32)
            hi16 = (d & -65536).view(torch.float32)
            return torch.stack([lo16, hi16], dim=-1).flatten(-2)
        v = unpack(g[..., 0, :]) + unpack(g[..., 1, :])
        return Feat(v.reshape(self.n, self.h, self.w, self.c).contiguous())

    def to_nchw(self) -> torch.Tensor:
        """Debug / boundary helper: dense NCHW copy (through the HIP layout kernel)."""
        if DISPATCH == "torch":
            return _tops().nhwc_to_nchw(self.view())
        out = torch.empty((self.n, self.c, self.h, self.w), device=self.device, dtype=torch.float32)
        L.check(L.load().prv2_nhwc_to_nchw(self.ptr, self.n, self.c, self.h, self.w, self.ld, out.data_ptr(),
                                           _stream()), "nhwc_to_nchw")
        return out

    @staticmethod
    def from_nchw(x: torch.Tensor, pad_to: int = 4) -> "Feat":
        _require_dev(x)
        x = x.contiguous()
        n, c, h, w = x.shape
        if DISPATCH == "torch" and pad_to == 4:
            y = _tops().nchw_to_nhwc(x)
            return _feat_of(y, c)
        f = Feat.alloc(n, h, w, c, x.device, pad_to)
        L.check(L.load().prv2_nchw_to_nhwc(x.data_ptr(), n, c, h, w, f.ptr, f.ld, _stream()), "nchw_to_nhwc")
        return f


@dataclass
class ConvW:
    """A packed convolution / linear weight (prv2_pack_conv_weight layout) + its bias."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    cout: int
    cin: int
    kh: int
    kw: int
    stride: int = 1
    pad: int = 0
    convt_k: int = 0
    prec: int = PREC_F32
    same_pad: bool = False  # timm Conv2dSame / TensorFlow "SAME" padding (prv2_conv_desc.same_pad)


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, pad: Optional[int] = None,
              convt_k: int = 0, bn_scale: Optional[torch.Tensor] = None, prec: int = PREC_F32, device=None,
              same_pad: bool = False) -> ConvW:
    """weight: PyTorch layout [cout, cin, kh, kw] / [cout, cin] (Linear) / [cin, cout, k, k] (ConvTranspose2d)."""
    lib = L.load()
    device = device or weight.device
    w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None]
    if convt_k:
        cin, cout, kh, kw = w.shape
    else:
        cout, cin, kh, kw = w.shape
    if pad is None:
        pad = kh // 2 if not convt_k else 0
    sc = bn_scale.detach().to(device=device, dtype=torch.float32).contiguous() if bn_scale is not None else None
    if DISPATCH == "torch":
        packed = _tops().pack_conv_weight(w, sc, convt_k, prec)
    else:
        nbytes = lib.prv2_packed_weight_bytes(cout, cin, kh, kw, convt_k, prec)
        packed = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
        L.check(lib.prv2_pack_conv_weight(w.data_ptr(), _ptr(sc), packed.data_ptr(), cout, cin, kh, kw, convt_k, prec,
                                          _stream()), "pack_conv_weight")
    b = bias.detach().to(device=device, dtype=torch.float32).contiguous() if bias is not None else None
    return ConvW(packed, b, cout, cin, kh, kw, convt_k if convt_k else stride, pad, convt_k, prec, same_pad)


@dataclass
class ConvWF6:
    """A 256-column 3x3 conv packed for the fp16 + fp6 kernel (prv2_pack_conv3x3_f6_weight) + its bias and power-of-two scales."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    cout: int
    cin: int
    w_scale: float          # the weights were multiplied by this before the fp16 / fp6 split (max |w w_scale| in [1, 2))
    x_scale: float = 1.0    # what the loader multiplies the activations with (calibration: ``range``)
    range: Optional[torch.Tensor] = None  # device uint32[1]: float bits of the largest |relu(x) x_scale| any launch saw


F16F6 = os.environ.get("PRV2_F16F6", "0") == "1"  # the fp16 + fp6 arithmetic for the layers that have a kernel for it (default: bf16x3)
F6_GATE = os.environ.get("PRV2_F6_GATE", "1") != "0"  # A/B switch of the mode's stage 2: the GatedConvUnit tail kernel (conv3x3_c256_gate_f6_kernel)


class F6Range:
    """The fp16 range guard of the fp16 + fp6 layers.  Every packed layer owns one word of a per-device table; its kernel raises the word
    to the float bits of the largest |relu(x) x_scale| it saw (prv2_conv3x3_f6, ``range_word``).  After a frame (one small D2H behind the
    frame's own synchronisation) ``check`` moves each layer's power-of-two ``x_scale`` so that its largest input sits near 2^8 -- 2^8 of
    head room below fp16's 65504, 22 binades of full-precision values below it -- and tells the caller whether the frame it just
    computed has to be recomputed: a layer saw values beyond fp16's range (its fp16 part saturated: finite but fp6-grade) or its whole
    input sat below 2^-10 (fp16 subnormals)."""
    SLOTS = 256
    moved = False
    TARGET_LOG2, LOW_LOG2, HIGH = 8, -10, 65504.0
    _tables: dict = {}

    @staticmethod
    def _key(device) -> str:
        """one key per physical device: 'cuda' (a config default) and 'cuda:0' (a tensor's .device) are the same table"""
        d = torch.device(device)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        return str(d)

    @classmethod
    def slot(cls, device, cw) -> torch.Tensor:
        import weakref
        device = torch.device(cls._key(device))
        ent = cls._tables.setdefault(str(device), dict(table=torch.zeros(cls.SLOTS, dtype=torch.int32, device=device), layers=[]))
        ent["layers"] = [(i, r) for i, r in ent["layers"] if r() is not None]
        used = {i for i, _ in ent["layers"]}
        i = next((k for k in range(cls.SLOTS) if k not in used), None)
        if i is None:
            raise RuntimeError(f"F6Range: all {cls.SLOTS} range words of {device} are taken by live fp16 + fp6 layers")
        ent["layers"].append((i, weakref.ref(cw)))
        ent["table"][i] = 0
        return ent["table"][i:i + 1]

    @classmethod
    def active(cls, device) -> bool:
        ent = cls._tables.get(cls._key(device))
        return bool(ent and any(r() is not None for _, r in ent["layers"]))

    @classmethod
    def check(cls, device, reduce=None) -> list:
        """-> [(layer, seen maximum, old x_scale)] of the layers whose frame must be recomputed (their x_scale is already moved); the
        table is cleared for the next frame.  ``reduce(table)``: an in-place MAX all-reduce over the ranks of a patch-sharded frame (the words are
        float bits of non-negative values: integer order == float order), so that every rank takes the same decision.
        ``cls.moved`` tells the caller whether ANY x_scale changed (captured hipGraphs carry the old scales as kernel arguments: the
        caller drops them, otherwise the next measurement would be taken under a scale this table no longer knows)"""
        cls.moved = False
        ent = cls._tables.get(cls._key(device))
        if not ent:
            return []
        if reduce is not None:
            reduce(ent["table"])
        seen = ent["table"].to("cpu", copy=True).view(torch.float32)  # (synchronises the current stream)
        ent["table"].zero_()
        redo = []
        for i, r in ent["layers"]:
            cw = r()
            m = float(seen[i]) if cw is not None else 0.0
            if m <= 0.0 or not math.isfinite(m):
                if cw is not None and not math.isfinite(m) and m != 0.0:
                    redo.append((cw, m, cw.x_scale))  # (inf / nan inputs: nothing a scale can do -- reported)
                continue
            e = math.floor(math.log2(m))
            if m > cls.HIGH or e < cls.LOW_LOG2:
                redo.append((cw, m, cw.x_scale))
            if m > cls.HIGH / 4 or e < cls.LOW_LOG2 + 6:  # (move early: two binades before the upper, six before the lower limit)
                cw.x_scale = cw.x_scale * 2.0 ** (cls.TARGET_LOG2 - e)
                cls.moved = True
        return redo

    @classmethod
    def clear(cls, device) -> None:
        """forget what the launches so far saw (callers that run the fp16 + fp6 layers outside a guarded frame)"""
        ent = cls._tables.get(cls._key(device))
        if ent:
            ent["table"].zero_()


def pack_conv3x3_f6(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, device=None) -> ConvWF6:
    """weight [256, cin, 3, 3] (cin % 64 == 0) -> the fragment-major fp16 + fp6 image of csrc/conv3x3_f6.hip."""
    lib = L.load()
    device = device or weight.device
    w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
    cout, cin, kh, kw = w.shape
    nbytes = lib.prv2_conv3x3_f6_weight_bytes(cout, cin)
    assert (kh, kw) == (3, 3) and nbytes > 0, (cout, cin, kh, kw)
    amax = float(w.abs().max())
    w_scale = 2.0 ** -math.floor(math.log2(amax)) if amax > 0 else 1.0
    packed = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
    if DISPATCH == "torch":
        _tops().pack_conv3x3_f6_weight(w, w_scale, packed)
    else:
        L.check(lib.prv2_pack_conv3x3_f6_weight(w.data_ptr(), w_scale, packed.data_ptr(), cout, cin, _stream()), "pack_conv3x3_f6_weight")
    b = bias.detach().to(device=device, dtype=torch.float32).contiguous() if bias is not None else None
    cw = ConvWF6(packed, b, cout, cin, w_scale, 1.0, None)
    cw.range = F6Range.slot(device, cw)
    return cw


def conv3x3_f6_supported(x: Feat, cout: int, cin: int, allow_x2: bool = False) -> bool:
    """shape contract of the fp16 + fp6 kernels (a property of the layer, never of the batch); ``allow_x2``: the gate-tail kernel also takes pre-split input"""
    if (getattr(x, "x2", False) and not allow_x2) or x.c != cin or x.ld % 4:
        return False
    d = L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cin, cout=cout, kh=3, kw=3, stride=1, pad=1, ldx=x.ld, ldy=roundup(cout, 4), x_bstride=0, y_bstride=0,
                   relu_in=0, act=ACT_NONE, convt_k=0, ld_mul=0, ld_res=0, ld_res2=0, prec=L.PREC_F16F6, force_generic=0, ln_eps=1e-6, part=0,
                   same_pad=0, fmt=0)
    return bool(L.load().prv2_conv3x3_f6_supported(C.byref(d)))


def conv3x3_f6(x: Feat, cw: ConvWF6, out: Optional[Feat] = None, *, relu_in: bool = False, res: Optional[Feat] = None) -> Feat:
    """y = conv3x3(relu?(x)) + bias (+ res) in the fp16 + fp6 arithmetic (include/prv2.h::prv2_conv3x3_f6); ``out`` may be X2."""
    assert x.c == cw.cin and not getattr(x, "x2", False)
    if out is None:
        out = Feat.alloc(x.n, x.h, x.w, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, cw.cout)
    assert res is None or ((res.n, res.h, res.w, res.c) == (out.n, out.h, out.w, out.c) and not res.x2)
    d = L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=3, kw=3, stride=1, pad=1, ldx=x.ld, ldy=out.ld, x_bstride=0, y_bstride=0,
                   relu_in=int(relu_in), act=ACT_NONE, convt_k=0, ld_mul=0, ld_res=res.ld if res is not None else 0, ld_res2=0, prec=L.PREC_F16F6,
                   force_generic=0, ln_eps=1e-6, part=0, same_pad=0, fmt=L.FMT_Y_X2 if out.x2 else 0)
    out_scale = 1.0 / (cw.x_scale * cw.w_scale)

    def call():
        if DISPATCH == "torch":
            _tops().conv3x3_f6(x.view(), cw.w, cw.bias, res.view() if res is not None else None, relu_in, cw.x_scale, out_scale, cw.range, out.raw(), d.fmt)
            return
        L.check(L.load().prv2_conv3x3_f6(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), _ptr(res), cw.x_scale, out_scale, _ptr(cw.range), out.ptr, _stream()),
                "conv3x3_f6")

    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), 2.0 * x.n * x.h * x.w * cw.cout * cw.cin * 9, call,
                    shape=f"{cw.cin}->{cw.cout} k3s1 {x.n}x{x.h}x{x.w}")
    return out


def conv_out_hw(cw: ConvW, h: int, w: int):
    if cw.convt_k:
        return h * cw.convt_k, w * cw.convt_k
    if cw.same_pad:
        return -(-h // cw.stride), -(-w // cw.stride)
    return (h + 2 * cw.pad - cw.kh) // cw.stride + 1, (w + 2 * cw.pad - cw.kw) // cw.stride + 1


def _c256(x: Feat, cw: ConvW) -> bool:
    """does the library run this conv on its 256-channel 3x3 kernel (which fuses the LayerNorm at that width)?"""
    if x.c != cw.cin or x.ld % 4 or cw.cout != 256 or os.environ.get("PRV2_NO_C256"):
        return False
    return bool(L.load().prv2_conv3x3_ln_gate_supported(C.byref(_gate_desc(x, cw, roundup(cw.cout, 4), False, ACT_NONE, None, None, 1e-6))))


def conv2d(x: Feat, cw: ConvW, out: Optional[Feat] = None, *, relu_in: bool = False, act: int = ACT_NONE,
           gamma: Optional[torch.Tensor] = None, mul: Optional[Feat] = None, res: Optional[Feat] = None,
           res2: Optional[Feat] = None, x_bstride: int = 0, force_generic: bool = False, ln=None,
           ln_eps: float = 1e-6, algo: Optional[float] = None) -> Feat:
    """y = epilogue(conv(x)); see include/prv2.h::prv2_conv2d.  ``ln`` = (weight, bias) of a channels-first
    LayerNorm applied between the bias and the activation (fused when cout <= 128, else a separate row-LN pass).
    ``algo``: the reference graph's FLOPs for this step when they differ from the executed ones (Profiler.launch)."""
    if ln is not None and cw.cout > 128 and not (gamma is None and mul is None and res2 is None and not force_generic and not x_bstride
                                                 and _c256(x, cw)):
        y = conv2d(x, cw, out, relu_in=relu_in, x_bstride=x_bstride, force_generic=force_generic)
        assert gamma is None and mul is None and res is None and res2 is None
        return layernorm_feat(y, ln[0], ln[1], ln_eps, act)
    assert x.c == cw.cin, (x.c, cw.cin)
    oh, ow = conv_out_hw(cw, x.h, x.w)
    if out is None:
        out = Feat.alloc(x.n, oh, ow, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, oh, ow, cw.cout), ((out.n, out.h, out.w, out.c), (x.n, oh, ow, cw.cout))
    d = L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad,
                   ldx=x.ld, ldy=out.ld, x_bstride=x_bstride, y_bstride=0, relu_in=int(relu_in), act=act,
                   convt_k=cw.convt_k, ld_mul=mul.ld if mul is not None else 0, ld_res=res.ld if res is not None else 0,
                   ld_res2=res2.ld if res2 is not None else 0, prec=cw.prec, force_generic=int(force_generic),
                   ln_eps=ln_eps, part=0, same_pad=int(cw.same_pad), fmt=0)
    for aux in (mul, res, res2):
        if aux is not None:
            assert (aux.n, aux.h, aux.w, aux.c) == (out.n, out.h, out.w, out.c) and not aux.x2
    assert not getattr(x, "x2", False), "conv2d reads fp32 activations (X2 inputs: conv3x3_ln_gate)"
    if out.x2:  # pre-split output: written by the 256-column conv without LayerNorm (GatedConvUnit.conv -> the unit's concat buffer)
        assert ln is None and gamma is None and mul is None and res2 is None and not force_generic and not x_bstride and _c256(x, cw)
        d.fmt = L.FMT_Y_X2
    ncols = cw.cout * (cw.convt_k ** 2 if cw.convt_k else 1)
    taps = 1 if cw.convt_k else cw.kh * cw.kw
    m_rows = x.n * (x.h * x.w if cw.convt_k else oh * ow)
    halo = (cw.kh == 3 and cw.kw == 3 and cw.stride == 1 and (cw.pad == 1 or cw.same_pad) and not cw.convt_k and x.w >= 24 and x.h >= 4
            and not force_generic)  # only for the f32-mode strip split below; kernel names come from prv2_last_kernel()
    via_torch = DISPATCH == "torch"

    def call():
        if via_torch:
            v = lambda f: None if f is None else f.view()  # noqa: E731
            _tops().conv2d(x.view(), cw.w, cw.bias, cw.cout, cw.kh, cw.kw, cw.stride, cw.pad, act, relu_in,
                           ln[0] if ln is not None else None, ln[1] if ln is not None else None, gamma, v(mul), v(res), v(res2),
                           cw.convt_k, cw.prec, ln_eps, cw.same_pad, out.raw(), d.fmt, force_generic, d.part)
            return
        L.check(L.load().prv2_conv2d(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), _ptr(ln[0]) if ln is not None else None,
                                     _ptr(ln[1]) if ln is not None else None, _ptr(gamma), _ptr(mul), _ptr(res), _ptr(res2),
                                     out.ptr, _stream()), "conv2d")

    shape = f"{cw.cin}->{cw.cout} k{cw.kh}s{cw.stride}{'T' if cw.convt_k else ''} {x.n}x{x.h}x{x.w}"
    tag = lambda: L.load().prv2_last_kernel().decode()  # noqa: E731  (the kernel the library dispatched this call to)
    rem = x.w % 32
    if halo and PROFILER.enabled and 0 < rem <= 8 and x.w >= 64 and cw.prec == PREC_F32:
        # f32 mode: the library runs this conv as 32-pixel tiles + a remainder strip on the generic kernel (csrc/igemm.hip;
        # the bf16 modes do both in one launch); when launches are timed the two kernels are issued and accounted separately
        full = 2.0 * m_rows * ncols * cw.cin * taps
        d.part = 1
        PROFILER.launch(tag, full * (x.w - rem) / x.w, call, shape=shape)
        d.part = 2
        PROFILER.launch(tag, full * rem / x.w, call, shape=shape + f" strip{rem}")
        return out
    PROFILER.launch(tag, 2.0 * m_rows * ncols * cw.cin * taps, call, shape=shape, algo=algo)
    return out


UPS_FUSION = os.environ.get("PRV2_UPS_FUSION", "1") != "0"  # A/B and test switch: bilinear upsample fused into the consumer conv's loader


def _ups_desc(x: Feat, u: Feat, cw: ConvW, out_ld: int, act: int, res_ld: int, ln_eps: float):
    d = L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad, ldx=x.ld, ldy=out_ld,
                   x_bstride=0, y_bstride=0, relu_in=0, act=act, convt_k=cw.convt_k, ld_mul=0, ld_res=res_ld, ld_res2=0, prec=cw.prec,
                   force_generic=0, ln_eps=ln_eps, part=0, same_pad=int(cw.same_pad), fmt=0)
    us = L.UpsSrc(x=u.ptr, h=u.h, w=u.w, ld=u.ld, channels=u.c, bstride=0)
    return d, us


class UpsOnly:
    """the ``x`` of ``conv2d_ups`` when EVERY input channel comes from the upsampled source (u.c == cin): only the output size"""

    def __init__(self, u: Feat, h: int, w: int):
        self.n, self.h, self.w, self.c, self.ld, self.ptr, self.device = u.n, h, w, u.c, u.ld, u.ptr, u.device


def conv2d_ups_supported(x, u: Feat, cw: ConvW) -> bool:
    """can ``conv2d_ups`` fuse the upsample of ``u`` (the first u.c input channels) into this 3x3 conv's loader?  A property of the
    layer (channels, per-image size, arithmetic mode) -- never of the batch."""
    if not UPS_FUSION or type(x) not in (Feat, UpsOnly) or type(u) is not Feat or x.n != u.n or (x.h, x.w) == (u.h, u.w):
        return False
    d, us = _ups_desc(x, u, cw, roundup(cw.cout, 4), ACT_NONE, 0, 1e-6)
    return bool(L.load().prv2_conv2d_ups_supported(C.byref(d), C.byref(us)))


UPCONV = os.environ.get("PRV2_UPCONV", "1") != "0"  # A/B and test switch: 3x3 convs of an upsampled tensor computed at the low resolution


def upconv3x3_supported(u: Feat, h: int, w: int, cw: ConvW) -> bool:
    """can ``upconv3x3`` take conv3x3(bilinear_align_corners(u -> h x w); cw)?  A property of the layer (channels, per-image sizes, mode)."""
    if not UPCONV or type(u) is not Feat or (cw.kh, cw.kw, cw.stride, cw.pad, cw.convt_k) != (3, 3, 1, 1, 0) or cw.cin != u.c or cw.same_pad:
        return False
    us = L.UpsSrc(x=u.ptr, h=u.h, w=u.w, ld=u.ld, channels=u.c, bstride=0)
    return bool(L.load().prv2_upconv3x3_supported(C.byref(us), u.n, h, w, cw.cout, cw.prec))


def upconv3x3(u: Feat, h: int, w: int, cw: ConvW, out: Optional[Feat] = None, *, act: int = ACT_NONE, bias: bool = True, add: Optional[Feat] = None) -> Feat:
    """act(conv3x3(bilinear_align_corners(u -> h x w); cw) + bias) computed at u's resolution (include/prv2.h::prv2_upconv3x3): nine tap
    GEMMs on the low-resolution grid, then the four interpolation corners of every tap gathered per output pixel -- 2.3x fewer matrix
    operations than ``conv2d_ups``.  fp32-grade, not bit-identical to it (summation order).  ``add`` [n, h, w, cout]: a pre-activation
    addend -- the direct conv over the REST of a concat input [up(u) | rest] with the other weight columns (it may be ``out`` itself)."""
    assert cw.cin == u.c
    if out is None:
        out = Feat.alloc(u.n, h, w, cw.cout, u.device)
    assert (out.n, out.h, out.w, out.c) == (u.n, h, w, cw.cout)
    b = cw.bias if bias else None

    def call():
        if DISPATCH == "torch":
            _tops().upconv3x3(u.view(), cw.w, b, cw.cout, h, w, act, cw.prec, out.view(), add.view() if add is not None else None)
            return
        us = L.UpsSrc(x=u.ptr, h=u.h, w=u.w, ld=u.ld, channels=u.c, bstride=0)
        L.check(L.load().prv2_upconv3x3(C.byref(us), cw.w.data_ptr(), _ptr(b), _ptr(add), add.ld if add is not None else 0, u.n, h, w, cw.cout, act, cw.prec,
                                        out.ptr, out.ld, 0, _stream()), "upconv3x3")

    # executed: the tap GEMMs over every tile's 192-pixel source footprint (16 x 28 output tiles, 32-channel passes); algo: the reference
    # graph's nine taps at the OUTPUT resolution
    wide = (u.h - 1) * 2 <= (h - 1) and (u.w - 1) * 2 <= (w - 1)  # (csrc/upconv.hip: 16 x 28 output tiles up to a source step of 1/2, else 14 x 24)
    tiles = -(-h // 16) * -(-w // 28) if wide else -(-h // 14) * -(-w // 24)
    shape = f"{cw.cin}->{cw.cout} k3s1 {u.n}x{h}x{w} (lowres {u.c}ch {u.h}x{u.w})"
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), 2.0 * u.n * tiles * 192 * 9 * cw.cin * roundup(cw.cout, 32), call, shape=shape,
                    algo=2.0 * u.n * h * w * cw.cout * cw.cin * 9)
    return out


UPCONV5 = os.environ.get("PRV2_UPCONV5", "1") != "0"  # A/B and test switch: output_conv2[0] o output_conv1 o interpolate as one 5x5 conv at the source resolution


def compose_upconv5x5(w1: torch.Tensor, b1: torch.Tensor, tap_bias: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor], device, prec) -> dict:
    """Host side (float64) of ``upconv5x5``: two back-to-back 3x3 convs -- w1 [m, ci, 3, 3] + bias b1 [m] (+ ``tap_bias`` [9, m]: what each tap of the
    first conv adds where it lies inside the image -- the folded bias of an upstream 1x1, fusion.py), w2 [co, m, 3, 3] + b2 -- as
    dict(w5 = packed 5x5 composite, bias_map [5, 5, co], edge = packed 1x1 weights [28 co, ci] of the ring fix).  The algebra:
    tools/studies/composite5x5_ring.py (include/prv2.h::prv2_upconv5x5)."""
    import torch.nn.functional as F
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    w1, w2 = d(w1), d(w2)
    m, ci = w1.shape[:2]
    co = w2.shape[0]
    b1 = d(b1) if b1 is not None else torch.zeros(m, dtype=torch.float64)
    b2 = d(b2) if b2 is not None else torch.zeros(co, dtype=torch.float64)
    tap_bias = d(tap_bias) if tap_bias is not None else torch.zeros(9, m, dtype=torch.float64)
    weff = torch.zeros(co, ci, 5, 5, dtype=torch.float64)
    for y2 in range(3):
        for x2 in range(3):
            weff[:, :, y2:y2 + 3, x2:x2 + 3] += torch.einsum("om,miyx->oiyx", w2[:, :, y2, x2], w1)
    inside = torch.ones(1, 1, 8, 8, dtype=torch.float64)
    tb = b1.view(1, m, 1, 1) + F.conv2d(inside, tap_bias.t().reshape(m, 1, 3, 3), padding=1)
    vb = F.conv2d(tb, w2, b2, padding=1)[0]
    cls = [0, 1, 3, 6, 7]  # rows / columns of the 8 x 8 grid standing for the classes 0, 1, interior, h - 2, h - 1
    bias_map = vb[:, cls][:, :, cls].permute(1, 2, 0).contiguous()

    def edge(k2sel, k1sel):
        we = torch.zeros(5, co, ci, dtype=torch.float64)
        for k2 in range(3):
            for k1 in range(3):
                we[k2 + k1] += k2sel(k2) @ k1sel(k1)
        return we
    zero = torch.zeros(co, ci, dtype=torch.float64)
    groups = [  # (five taps along the edge, corner term at the line's first position, at its last) per edge: top, bottom, left, right
        (edge(lambda k: w2[:, :, 0, k], lambda k: w1[:, :, 2, k]), w2[:, :, 0, 0] @ w1[:, :, 2, 2], w2[:, :, 0, 2] @ w1[:, :, 2, 0]),
        (edge(lambda k: w2[:, :, 2, k], lambda k: w1[:, :, 0, k]), w2[:, :, 2, 0] @ w1[:, :, 0, 2], w2[:, :, 2, 2] @ w1[:, :, 0, 0]),
        (edge(lambda k: w2[:, :, k, 0], lambda k: w1[:, :, k, 2]), zero, zero),
        (edge(lambda k: w2[:, :, k, 2], lambda k: w1[:, :, k, 0]), zero, zero)]
    wedge = torch.cat([torch.cat([we.reshape(5 * co, ci), c0, c1], 0) for we, c0, c1 in groups], 0)  # [28 co, ci]
    return dict(w5=pack_conv(weff.float(), None, pad=2, device=device, prec=prec), bias_map=bias_map.float().contiguous().to(device),
                edge=pack_conv(wedge.float(), None, device=device, prec=prec), cout=co)


def upconv5x5_supported(u: Feat, h: int, w: int, cw5: dict) -> bool:
    if not UPCONV5 or type(u) is not Feat or cw5["w5"].cin != u.c:
        return False
    us = L.UpsSrc(x=u.ptr, h=u.h, w=u.w, ld=u.ld, channels=u.c, bstride=0)
    return bool(L.load().prv2_upconv5x5_supported(C.byref(us), u.n, h, w, cw5["cout"], cw5["w5"].prec))


def upconv5x5(u: Feat, h: int, w: int, cw5: dict, out: Optional[Feat] = None, *, act: int = ACT_NONE) -> Feat:
    """act(conv3x3(conv3x3(bilinear_align_corners(u -> h x w); W1) + b1; W2) + b2) as ONE 5x5 conv at u's resolution + the bias classes + the
    fix of the one-pixel border ring (include/prv2.h::prv2_upconv5x5 / _lines / _ring; ``compose_upconv5x5``): the 3x3 pair's intermediate
    map is never formed.  fp32-grade, not bit-identical to the two-conv sequence."""
    cw, co = cw5["w5"], cw5["cout"]
    if out is None:
        out = Feat.alloc(u.n, h, w, co, u.device)
    assert (out.n, out.h, out.w, out.c) == (u.n, h, w, co) and not out.x2
    us = L.UpsSrc(x=u.ptr, h=u.h, w=u.w, ld=u.ld, channels=u.c, bstride=0)

    def main():
        if DISPATCH == "torch":
            _tops().upconv5x5(u.view(), cw.w, cw5["bias_map"], co, h, w, act, cw.prec, out.view())
            return
        L.check(L.load().prv2_upconv5x5(C.byref(us), cw.w.data_ptr(), cw5["bias_map"].data_ptr(), u.n, h, w, co, act, cw.prec, out.ptr, out.ld, 0, _stream()), "upconv5x5")

    tiles = -(-h // 14) * -(-w // 24)  # (csrc/upconv5.hip: 14 x 24 output tiles, 192-pixel source footprint, five 160-column kernel-row passes)
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), 2.0 * u.n * tiles * 192 * 25 * cw.cin * 32, main,
                    shape=f"{cw.cin}->(128)->{co} k5 {u.n}x{h}x{w} (lowres {u.c}ch {u.h}x{u.w})", algo=cw5.get("algo_per_px", 0.0) * u.n * h * w)
    # the ring: the four border lines of up(u) at u's resolution -> tap GEMMs of the edges -> 1-D gather on the ring pixels
    npos = 2 * u.w + 2 * u.h
    if DISPATCH == "torch":
        lines = Feat(_tops().upconv5x5_lines(u.view(), h, w))
    else:
        lines = Feat(torch.empty((u.n, 1, npos, u.c), device=u.device, dtype=torch.float32))
        L.check(L.load().prv2_upconv5x5_lines(C.byref(us), u.n, h, w, lines.ptr, _stream()), "upconv5x5_lines")
    ge = conv2d(lines, cw5["edge"], algo=0.0)
    if DISPATCH == "torch":
        _tops().upconv5x5_ring_(out.view(), ge.view(), u.h, u.w, act)
    else:
        L.check(L.load().prv2_upconv5x5_ring(out.ptr, out.ld, 0, u.n, h, w, co, ge.ptr, ge.ld, u.h, u.w, act, _stream()), "upconv5x5_ring")
    return out


def conv2d_ups(x: Feat, u: Feat, cw: ConvW, out: Optional[Feat] = None, *, act: int = ACT_NONE, res: Optional[Feat] = None, ln=None,
               ln_eps: float = 1e-6) -> Feat:
    """3x3 conv over the VIRTUAL concat [bilinear_align_corners(u -> x.h x x.w) | x[..., u.c:]]: channels [0, u.c) are interpolated
    from the low-resolution ``u`` inside the conv's tile loader (include/prv2.h::prv2_conv2d_ups); ``x`` supplies the rest at
    their usual channel offsets (its first u.c channels are never read).  Bit-identical to upsample_bilinear(u, out=x.slice(0, u.c))
    followed by conv2d(x, cw)."""
    assert x.c == cw.cin and u.c <= x.c and (u.n, x.n) == (x.n, u.n)
    oh, ow = conv_out_hw(cw, x.h, x.w)
    if out is None:
        out = Feat.alloc(x.n, oh, ow, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, oh, ow, cw.cout)
    d, us = _ups_desc(x, u, cw, out.ld, act, res.ld if res is not None else 0, ln_eps)

    def call():
        if DISPATCH == "torch":
            _tops().conv3x3_ups(None if type(x) is UpsOnly else x.view(), u.view(), cw.w, cw.bias, cw.cout, x.h, x.w, act, ln[0] if ln is not None else None,
                                ln[1] if ln is not None else None, res.view() if res is not None else None, cw.prec, ln_eps, out.view())
            return
        L.check(L.load().prv2_conv2d_ups(C.byref(d), x.ptr, C.byref(us), cw.w.data_ptr(), _ptr(cw.bias), _ptr(ln[0]) if ln is not None else None,
                                         _ptr(ln[1]) if ln is not None else None, _ptr(res), out.ptr, _stream()), "conv2d_ups")

    shape = f"{cw.cin}->{cw.cout} k3s1 {x.n}x{x.h}x{x.w} (+up {u.c}ch {u.h}x{u.w})"
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), 2.0 * x.n * oh * ow * cw.cout * cw.cin * 9, call, shape=shape)
    return out


TAIL_FUSION = os.environ.get("PRV2_TAIL_FUSION", "1") != "0"  # A/B and test switch: [pred1 | pred2] tails written by the conv that fills the row


def _tail_desc(x: Feat, cw: ConvW, out: Feat, act: int, res_ld: int, ln_eps: float):
    return L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad, ldx=x.ld, ldy=out.ld,
                      x_bstride=0, y_bstride=0, relu_in=0, act=act, convt_k=cw.convt_k, ld_mul=0, ld_res=res_ld, ld_res2=0, prec=cw.prec,
                      force_generic=0, ln_eps=ln_eps, part=0, same_pad=int(cw.same_pad), fmt=0)


def conv2d_tail_supported(x: Feat, cw: ConvW, out: Feat) -> bool:
    """can ``conv2d_tail`` close the [.. | pred1 | pred2 | 0 | 0] row behind this conv's output slice?  (a property of the layer)"""
    if not TAIL_FUSION or not DIRECT_PLACEMENT or type(x) is not Feat or out.ld != out.c0 + cw.cout + 4 or (out.c0 + cw.cout) % 4:
        return False
    d = _tail_desc(x, cw, out, ACT_NONE, 0, 1e-6)
    return bool(L.load().prv2_conv2d_tail_supported(C.byref(d)))


def conv2d_tail(x: Feat, cw: ConvW, out: Feat, p1: Feat, p2: Feat, *, act: int = ACT_NONE, ln=None, ln_eps: float = 1e-6) -> Feat:
    """conv2d(x, cw, out, act=, ln=) that also writes (p1, p2, 0, 0) -- the two dense depth maps resized bilinear(align_corners) to
    the output size -- into the four channels behind ``out``'s slice (include/prv2.h::prv2_conv2d_tail); bit-identical to
    conv2d + depth_pair_fill"""
    assert x.c == cw.cin and (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, cw.cout) and out.ld == out.c0 + cw.cout + 4
    assert p1.c == 1 and p2.c == 1 and p1.ld == 1 and p2.ld == 1 and (p1.n, p1.h, p1.w) == (p2.n, p2.h, p2.w) and p1.n == x.n
    d = _tail_desc(x, cw, out, act, 0, ln_eps)

    def call():
        if DISPATCH == "torch":
            _tops().conv3x3_tail(x.view(), cw.w, cw.bias, cw.cout, act, ln[0] if ln is not None else None, ln[1] if ln is not None else None, None,
                                 p1.buf.view(p1.n, p1.h, p1.w), p2.buf.view(p2.n, p2.h, p2.w), cw.prec, ln_eps, out.view())
            return
        L.check(L.load().prv2_conv2d_tail(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), _ptr(ln[0]) if ln is not None else None,
                                          _ptr(ln[1]) if ln is not None else None, None, p1.ptr, p2.ptr, p1.h, p1.w, out.ptr, _stream()), "conv2d_tail")

    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), 2.0 * x.n * x.h * x.w * cw.cout * cw.cin * 9, call,
                    shape=f"{cw.cin}->{cw.cout} k3s1 {x.n}x{x.h}x{x.w} (+tail)")
    return out


GATE_FUSION = os.environ.get("PRV2_GATE_FUSION", "1") != "0"  # A/B and test switch: GatedConvUnit tail as one kernel
X2_FORMAT = os.environ.get("PRV2_X2", "1") != "0"  # A/B and test switch: the GatedConvUnit's concat buffer in the pre-split operand format
# widths F of a GatedConvUnit the library fuses (prv2_conv3x3_ln_gate); PRV2_GATE_CHANNELS=256 restricts them for A/B runs
GATE_CHANNELS = tuple(int(c) for c in os.environ.get("PRV2_GATE_CHANNELS", "32,128,256").split(","))


def pack_gate(weight: torch.Tensor) -> torch.Tensor:
    """fragment-major image of a C -> C 1x1 conv's weights (C = 32, 128, 256) for ``conv3x3_ln_gate`` (prv2_pack_gate_weight)"""
    w = weight.detach().to(torch.float32).reshape(weight.shape[0], -1).contiguous()
    c = w.shape[0]
    assert w.shape == (c, c) and c in GATE_CHANNELS and w.is_cuda
    if DISPATCH == "torch":
        return _tops().pack_gate_weight(w)
    dst = torch.empty(L.load().prv2_gate_weight_bytes(c) // 4, device=w.device, dtype=torch.float32)
    _require_dev(w)
    L.check(L.load().prv2_pack_gate_weight(w.data_ptr(), dst.data_ptr(), c, c, _stream()), "pack_gate_weight")
    return dst


def _gate_desc(x: Feat, cw: ConvW, out_ld: int, relu_in, act, mul, res, ln_eps):
    fmt = (L.FMT_X_X2 if getattr(x, "x2", False) else 0) | (L.FMT_MUL_X2 if mul is not None and mul.x2 else 0)
    return L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad,
                      ldx=x.ld, ldy=out_ld, x_bstride=0, y_bstride=0, relu_in=int(relu_in), act=act, convt_k=cw.convt_k,
                      ld_mul=mul.ld if mul is not None else 0, ld_res=res.ld if res is not None else 0, ld_res2=0, prec=cw.prec,
                      force_generic=0, ln_eps=ln_eps, part=0, same_pad=int(cw.same_pad), fmt=fmt)


def conv3x3_ln_gate_supported(x: Feat, cw: ConvW) -> bool:
    """shape contract of the fused kernel (layer shape per image only: the choice never depends on the batch)"""
    if not GATE_FUSION or x.c != cw.cin or x.ld % 4 or cw.cout not in GATE_CHANNELS:
        return False
    return bool(L.load().prv2_conv3x3_ln_gate_supported(C.byref(_gate_desc(x, cw, roundup(cw.cout, 4), False, ACT_NONE, None, None, 1e-6))))


def conv3x3_ln_gate(x: Feat, cw: ConvW, ln, gate_w: Optional[torch.Tensor], gate_bias: Optional[torch.Tensor], out: Optional[Feat] = None,
                    *, act: int = ACT_RELU, mul: Optional[Feat] = None, res: Optional[Feat] = None, relu_in: bool = False,
                    ln_eps: float = 1e-6, pre: Optional[Feat] = None, pre_cin: int = 0) -> Feat:
    """y = mul * sigmoid(conv1x1(act(LN(conv3x3(x) + b))) + gate_bias) (+ res) in one kernel (include/prv2.h::prv2_conv3x3_ln_gate);
    ``gate_w`` None: y = act(LN(conv3x3(x) + b)).  ``pre``: pre-LayerNorm addend [n, h, w, cout] -- the conv's coarse half from
    ``CoarseTaps.gather`` (prv2_conv3x3_ln_gate_pre), standing for ``pre_cin`` further input channels of the reference's conv."""
    if out is None:
        out = Feat.alloc(x.n, x.h, x.w, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, cw.cout) and x.c == cw.cin
    for aux in (mul, res):
        assert aux is None or (aux.n, aux.h, aux.w, aux.c) == (out.n, out.h, out.w, out.c)
    d = _gate_desc(x, cw, out.ld, relu_in, act, mul, res, ln_eps)
    assert not out.x2 and (res is None or not res.x2)
    flops = 2.0 * x.n * x.h * x.w * cw.cout * (cw.cin * 9 + (cw.cout if gate_w is not None else 0))
    if pre is not None:
        assert (pre.n, pre.h, pre.w, pre.c) == (out.n, out.h, out.w, out.c) and not pre.x2

    def call():
        if DISPATCH == "torch":
            r = lambda f: None if f is None else f.raw()  # noqa: E731
            _tops().conv3x3_ln_gate(x.raw(), cw.w, cw.bias, ln[0], ln[1], gate_w, gate_bias, r(mul), r(res), act, relu_in, cw.prec, ln_eps, out.view(),
                                    r(pre), d.fmt)
            return
        L.check(L.load().prv2_conv3x3_ln_gate_pre(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), _ptr(pre), pre.ld if pre is not None else 0, _ptr(ln[0]),
                                                  _ptr(ln[1]), _ptr(gate_w), _ptr(gate_bias), _ptr(mul), _ptr(res), out.ptr, _stream()), "conv3x3_ln_gate")

    coarse = f"(+{pre_cin} coarse)" if pre is not None else ""
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), flops, call,
                    shape=f"{cw.cin}{coarse}->{cw.cout}{'->' + str(cw.cout) + ' gate' if gate_w is not None else ''} k3s1 {x.n}x{x.h}x{x.w}",
                    algo=flops + 2.0 * x.n * x.h * x.w * cw.cout * pre_cin * 9 if pre is not None else None)
    return out


def conv3x3_ln_gate_f6(x: Feat, cw: ConvWF6, ln, gate_w: torch.Tensor, gate_bias: Optional[torch.Tensor], out: Optional[Feat] = None, *,
                       act: int = ACT_RELU, mul: Optional[Feat] = None, res: Optional[Feat] = None, ln_eps: float = 1e-6,
                       pre: Optional[Feat] = None, pre_cin: int = 0) -> Feat:
    """``conv3x3_ln_gate`` with the 3x3 conv in the fp16 + fp6 arithmetic (include/prv2.h::prv2_conv3x3_ln_gate_f6): x and mul in one format -- the unit's
    pre-split (X2) ``out`` / concat, or the fp32 concat; LayerNorm, gate GEMM (bf16x3) and final stage are the bf16x3 kernel's."""
    x2 = bool(getattr(x, "x2", False))
    assert x.c == cw.cin and (mul is None or bool(mul.x2) == x2)
    if out is None:
        out = Feat.alloc(x.n, x.h, x.w, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, cw.cout) and not out.x2 and (res is None or not res.x2)
    for aux in (mul, res, pre):
        assert aux is None or (aux.n, aux.h, aux.w, aux.c) == (out.n, out.h, out.w, out.c)
    d = L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=3, kw=3, stride=1, pad=1, ldx=x.ld, ldy=out.ld, x_bstride=0, y_bstride=0,
                   relu_in=0, act=act, convt_k=0, ld_mul=mul.ld if mul is not None else 0, ld_res=res.ld if res is not None else 0, ld_res2=0,
                   prec=L.PREC_F16F6, force_generic=0, ln_eps=ln_eps, part=0, same_pad=0,
                   fmt=(L.FMT_X_X2 | (L.FMT_MUL_X2 if mul is not None else 0)) if x2 else 0)
    out_scale = 1.0 / (cw.x_scale * cw.w_scale)
    flops = 2.0 * x.n * x.h * x.w * cw.cout * (cw.cin * 9 + cw.cout)

    def call():
        if DISPATCH == "torch":
            r = lambda f: None if f is None else f.raw()  # noqa: E731
            _tops().conv3x3_ln_gate_f6(x.raw(), cw.w, cw.bias, r(pre), ln[0], ln[1], gate_w, gate_bias, r(mul), r(res), act, ln_eps, cw.x_scale, out_scale,
                                       cw.range, out.view(), d.fmt)
            return
        L.check(L.load().prv2_conv3x3_ln_gate_f6(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), _ptr(pre), pre.ld if pre is not None else 0, _ptr(ln[0]),
                                                 _ptr(ln[1]), _ptr(gate_w), _ptr(gate_bias), _ptr(mul), _ptr(res), cw.x_scale, out_scale, _ptr(cw.range),
                                                 out.ptr, _stream()), "conv3x3_ln_gate_f6")

    coarse = f"(+{pre_cin} coarse)" if pre is not None else ""
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), flops, call, shape=f"{cw.cin}{coarse}->{cw.cout}->{cw.cout} gate k3s1 {x.n}x{x.h}x{x.w}",
                    algo=flops + 2.0 * x.n * x.h * x.w * cw.cout * pre_cin * 9 if pre is not None else None)
    return out


def linear(x2d: torch.Tensor, cw: ConvW, out: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
    """rows [M, K] @ W^T (+ epilogue) -> [M, N]; a 1x1 convolution over an M x 1 image."""
    M, K = x2d.shape
    xf = Feat(x2d.view(1, M, 1, K), cw.cin)
    if out is None:
        out = torch.empty((M, cw.cout), device=x2d.device, dtype=torch.float32)
    of = Feat(out.view(1, M, 1, out.shape[1]), cw.cout)
    for k in ("mul", "res", "res2"):
        if kw.get(k) is not None and not isinstance(kw[k], Feat):
            t = kw[k]
            kw[k] = Feat(t.view(1, M, 1, t.shape[1]), cw.cout)
    conv2d(xf, cw, of, **kw)
    return out


def conv2d_cout1(x: Feat, weight: torch.Tensor, bias: Optional[torch.Tensor], k: int, *, act: int = ACT_NONE,
                 scale: float = 1.0, res: Optional[torch.Tensor] = None, clamp0: bool = False,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Single-output-channel conv -> dense [n, 1, h, w] tensor (NCHW == NHWC for one channel)."""
    y = out if out is not None else torch.empty((x.n, 1, x.h, x.w), device=x.device, dtype=torch.float32)
    assert y.is_contiguous() and y.numel() == x.n * x.h * x.w
    def call():
        if DISPATCH == "torch":
            _tops().conv_cout1(x.view(), weight, bias, k, act, scale, res, clamp0, y)
            return
        L.check(L.load().prv2_conv2d_cout1(x.ptr, x.n, x.h, x.w, x.c, x.ld, weight.data_ptr(), k, _ptr(bias), act, scale, _ptr(res), int(clamp0),
                                           y.data_ptr(), _stream()), "conv2d_cout1")

    PROFILER.launch("conv_cout1_kernel", 2.0 * x.n * x.h * x.w * x.c * k * k, call)
    return y


def dwconv2d(x: Feat, w_tapmajor: torch.Tensor, bias: Optional[torch.Tensor], k: int, stride: int, relu=False, *,
             act: Optional[int] = None, same_pad: bool = False) -> Feat:
    """depthwise k x k; ``relu`` (bool) or any ``act``; ``same_pad``: timm Conv2dSame (output ceil(h / stride))"""
    act = (ACT_RELU if relu else ACT_NONE) if act is None else act
    if same_pad:
        oh, ow = -(-x.h // stride), -(-x.w // stride)
    else:
        oh = (x.h + 2 * (k // 2) - k) // stride + 1
        ow = (x.w + 2 * (k // 2) - k) // stride + 1
    if DISPATCH == "torch":
        box = []
        PROFILER.launch("dwconv_kernel", 2.0 * x.n * oh * ow * x.c * k * k,
                        lambda: box.append(_tops().dwconv2d(x.view(), w_tapmajor, bias, k, stride, act, same_pad)))
        y = box[0]
        return _feat_of(y, x.c)
    out = Feat.alloc(x.n, oh, ow, x.c, x.device)
    PROFILER.launch("dwconv_kernel", 2.0 * x.n * oh * ow * x.c * k * k,
                    lambda: L.check(L.load().prv2_dwconv2d_ex(x.ptr, x.n, x.h, x.w, x.c, x.ld, w_tapmajor.data_ptr(),
                                                              _ptr(bias), k, stride, act, int(same_pad), out.ptr, out.ld,
                                                              _stream()), "dwconv2d"))
    return out


def global_avgpool(x: Feat) -> torch.Tensor:
    """[n, c] mean over the pixels (the squeeze of timm's SqueezeExcite: x.mean((2, 3)))"""
    assert x.c % 4 == 0
    if DISPATCH == "torch":
        box = []
        PROFILER.launch_aux("global_avgpool", 4.0 * x.n * x.h * x.w * x.c, lambda: box.append(_tops().global_avgpool(x.view())), f"{x.c}ch {x.n}x{x.h}x{x.w}")
        return box[0]
    out = torch.empty((x.n, x.c), device=x.device, dtype=torch.float32)
    ws = torch.empty(L.load().prv2_global_avgpool_workspace_floats(x.n, x.h * x.w, x.c), device=x.device, dtype=torch.float32)
    PROFILER.launch_aux("global_avgpool", 4.0 * x.n * x.h * x.w * x.c,
                        lambda: L.check(L.load().prv2_global_avgpool(x.ptr, x.n, x.h * x.w, x.c, x.ld, out.data_ptr(), ws.data_ptr(),
                                                                      _stream()),
                                        "global_avgpool"), f"{x.c}ch {x.n}x{x.h}x{x.w}")
    return out


def se_gate(mean: torch.Tensor, w1: torch.Tensor, b1, w2t: torch.Tensor, b2) -> torch.Tensor:
    """sigmoid(W2 silu(W1 mean + b1) + b2) per image: timm SqueezeExcite's conv_reduce -> act -> conv_expand -> gate.
    w1 [cse, c]; w2t [cse, c] = conv_expand's weight transposed."""
    n, c = mean.shape
    cse = w1.shape[0]
    assert w1.shape == (cse, c) and w2t.shape == (cse, c) and w1.is_contiguous() and w2t.is_contiguous()
    if DISPATCH == "torch":
        box = []
        PROFILER.launch_aux("se_gate", 4.0 * (n * c + 2 * c * cse), lambda: box.append(_tops().se_gate(mean, w1, b1, w2t, b2)), f"{c}->{cse}->{c} x{n}")
        return box[0]
    g = torch.empty((n, c), device=mean.device, dtype=torch.float32)
    ws = torch.empty((n, cse), device=mean.device, dtype=torch.float32)
    PROFILER.launch_aux("se_gate", 4.0 * (n * c + 2 * c * cse),
                        lambda: L.check(L.load().prv2_se_gate(mean.data_ptr(), n, c, w1.data_ptr(), _ptr(b1), cse, w2t.data_ptr(),
                                                               _ptr(b2), g.data_ptr(), ws.data_ptr(), _stream()), "se_gate"),
                        f"{c}->{cse}->{c} x{n}")
    return g


def channel_scale_(x: Feat, s: torch.Tensor) -> Feat:
    """x *= s[n, c] in place (the excite of SqueezeExcite)"""
    assert s.shape == (x.n, x.c) and s.is_contiguous() and x.c % 4 == 0
    PROFILER.launch_aux("channel_scale", 8.0 * x.n * x.h * x.w * x.c,
                        (lambda: _tops().channel_scale_(x.view(), s)) if DISPATCH == "torch" else
                        (lambda: L.check(L.load().prv2_channel_scale(x.ptr, x.n, x.h * x.w, x.c, x.ld, s.data_ptr(), _stream()), "channel_scale")),
                        f"{x.c}ch {x.n}x{x.h}x{x.w}")
    return x


def layernorm_rows(x: torch.Tensor, rows: int, c: int, ldx: int, weight, bias, eps: float, act: int, y: torch.Tensor,
                   ldy: int, x_off: int = 0, y_off: int = 0):
    if DISPATCH == "torch":
        _tops().layernorm(torch.as_strided(x, (rows, c), (ldx, 1), x.storage_offset() + x_off), weight, bias, eps, act,
                          torch.as_strided(y, (rows, c), (ldy, 1), y.storage_offset() + y_off))
        return
    L.check(L.load().prv2_layernorm(x.data_ptr() + 4 * x_off, rows, c, ldx, weight.data_ptr(), bias.data_ptr(), eps,
                                    act, y.data_ptr() + 4 * y_off, ldy, _stream()), "layernorm")


def layernorm_feat(x: Feat, weight, bias, eps: float = 1e-6, act: int = ACT_NONE, out: Optional[Feat] = None) -> Feat:
    """channels-first LayerNorm of the reference == row LayerNorm in NHWC (convs.py:21-29)."""
    if out is None:
        out = x
    if DISPATCH == "torch":
        rows = x.n * x.h * x.w
        PROFILER.launch_aux("layernorm", 8.0 * rows * x.c,
                            lambda: _tops().layernorm(x.view().reshape(rows, x.c) if x.ld == x.c else torch.as_strided(x.buf, (rows, x.c), (x.ld, 1), x.buf.storage_offset() + x.c0),
                                                      weight, bias, eps, act,
                                                      torch.as_strided(out.buf, (rows, out.c), (out.ld, 1), out.buf.storage_offset() + out.c0)),
                            f"{x.c}ch {x.n}x{x.h}x{x.w}")
        return out
    PROFILER.launch_aux("layernorm", 8.0 * x.n * x.h * x.w * x.c,
                        lambda: L.check(L.load().prv2_layernorm(x.ptr, x.n * x.h * x.w, x.c, x.ld, weight.data_ptr(),
                                                                 bias.data_ptr(), eps, act, out.ptr, out.ld, _stream()),
                                        "layernorm"), f"{x.c}ch {x.n}x{x.h}x{x.w}")
    return out


# ---- split-swizzled ("ss") operands of the large ViT linears (csrc/gemm_ss.hip; include/prv2.h "Split-swizzled") --------------
# An ss tensor is carried as a float32 [rows, C] container (same bytes per element as fp32); only the kernels interpret it.
SS_DISABLED = False  # A/B and test switch: keep the ViT blocks on the fp32-operand kernels
SS_MIN_ROWS = 512   # token rows from which the ViT blocks run on the pre-split path (bit-identical to the fp32-operand path;
                    # measured per linear vs gemm16: +25..37 % at 14 k rows, +20 % at 4 k, +30 % / 0 / -8 % (qkv, fc1 / proj / fc2) at
                    # 1037: profiles/r02_gemm_ss_bench.txt)


def split_ss(x2d: torch.Tensor) -> torch.Tensor:
    M, K = x2d.shape
    _require_dev(x2d)
    if DISPATCH == "torch":
        return _tops().split_ss(x2d)
    out = torch.empty((M, K), device=x2d.device, dtype=torch.float32)
    L.check(L.load().prv2_split_ss(x2d.data_ptr(), M, K, x2d.stride(0), out.data_ptr(), _stream()), "split_ss")
    return out


def layernorm_ss(x: torch.Tensor, rows: int, c: int, ldx: int, weight, bias, eps: float, y_ss: torch.Tensor):
    if DISPATCH == "torch":
        return _tops().layernorm_ss(torch.as_strided(x, (rows, c), (ldx, 1)), weight, bias, eps, y_ss)
    L.check(L.load().prv2_layernorm_ss(x.data_ptr(), rows, c, ldx, weight.data_ptr(), bias.data_ptr(), eps, y_ss.data_ptr(),
                                       _stream()), "layernorm_ss")


def gemm_ss(a_ss: torch.Tensor, cw: ConvW, out: Optional[torch.Tensor] = None, *, out_ss: bool = False, act: int = ACT_NONE,
            gamma: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rows (split-swizzled) [M, K] @ W^T (+ bias, act, gamma, residual) -> fp32 rows, or split-swizzled rows (out_ss)"""
    M, K = a_ss.shape
    assert cw.prec == L.PREC_BF16X3 and cw.kh == 1 and cw.kw == 1 and cw.cin == K and K % 32 == 0, (cw.prec, cw.cin, K)
    if out is None:
        out = torch.empty((M, cw.cout), device=a_ss.device, dtype=torch.float32)
    lib = L.load()

    def call():
        if DISPATCH == "torch":
            _tops().gemm_ss(a_ss, cw.w, cw.cout, cw.bias, gamma, res, act, out_ss, out)
            return
        L.check(lib.prv2_gemm_ss(a_ss.data_ptr(), M, K, cw.w.data_ptr(), cw.cout, _ptr(cw.bias), _ptr(gamma), _ptr(res),
                                 res.stride(0) if res is not None else 0, act, None if out_ss else out.data_ptr(), out.stride(0),
                                 out.data_ptr() if out_ss else None, _stream()), "gemm_ss")
    PROFILER.launch(lambda: lib.prv2_last_kernel().decode(), 2.0 * M * K * cw.cout, call, shape=f"{K}->{cw.cout} k1s1 1x{M}x1")
    return out


QKV_SS = os.environ.get("PRV2_QKV_SS", "1") != "0"  # A/B and test switch: the attention block without the qkv_split pre-pass
Q_SCALE = float(np.float32(0.125) * np.float32(1.4426950408889634))  # hd^-0.5 log2 e at head_dim 64: the float32 product qkv_split_kernel multiplies with


def gemm_ss_qkv(a_ss: torch.Tensor, cw: ConvW, heads: int) -> torch.Tensor:
    """the qkv Linear of an attention block on split-swizzled rows -> split-swizzled [q * hd^-0.5 log2 e | k | v] rows, the operand of
    ``attention_qkv_ss`` (include/prv2.h::prv2_gemm_ss_qkv)"""
    M, K = a_ss.shape
    assert cw.prec == L.PREC_BF16X3 and cw.kh == 1 and cw.kw == 1 and cw.cin == K and K % 32 == 0 and cw.cout == 3 * heads * 64, (cw.prec, cw.cin, K, cw.cout)
    lib = L.load()
    q_scale = Q_SCALE
    res = []

    def call():
        if DISPATCH == "torch":
            res.append(_tops().gemm_ss_qkv(a_ss, cw.w, cw.cout, cw.bias, heads * 64, q_scale))
            return
        out = torch.empty((M, cw.cout), device=a_ss.device, dtype=torch.float32)
        L.check(lib.prv2_gemm_ss_qkv(a_ss.data_ptr(), M, K, cw.w.data_ptr(), cw.cout, _ptr(cw.bias), heads * 64, q_scale, out.data_ptr(), _stream()), "gemm_ss_qkv")
        res.append(out)
    PROFILER.launch(lambda: lib.prv2_last_kernel().decode(), 2.0 * M * K * cw.cout, call, shape=f"{K}->{cw.cout} k1s1 1x{M}x1")
    return res[0]


def attention_qkv_ss(qkv_ss: torch.Tensor, b: int, ntok: int, heads: int, bias=None, out_ss: bool = True) -> torch.Tensor:
    """softmax(q k^T + bias) v on ``gemm_ss_qkv``'s rows (bf16x3); bit-equal to ``attention`` on the fp32 rows of the same Linear"""
    lib = L.load()
    image = isinstance(bias, AttnBiasImage)
    if image:
        assert (bias.heads, bias.ntok) == (heads, ntok)
        bias_t, ld_bias = bias.image, -1
    elif bias is not None:
        _require_dev(bias)
        assert bias.is_contiguous() and bias.shape[0] == heads and bias.shape[1] == ntok
        bias_t, ld_bias = bias, bias.shape[2]
    else:
        bias_t, ld_bias = None, 0
    res = []

    def call():
        if DISPATCH == "torch":
            res.append(_tops().attention_qkv_ss(qkv_ss, b, ntok, heads, bias_t, image, out_ss))
            return
        out = torch.empty((b * ntok, heads * 64), device=qkv_ss.device, dtype=torch.float32)
        L.check(lib.prv2_attention_qkv_ss(qkv_ss.data_ptr(), b, ntok, heads, 64, _ptr(bias_t), ld_bias, None if out_ss else out.data_ptr(),
                                          out.data_ptr() if out_ss else None, _stream()), "attention_qkv_ss")
        res.append(out)
    PROFILER.launch("attention_qkvss_kernel", 4.0 * b * heads * ntok * ntok * 64, call)
    return res[0]


def patchify(img: Feat, p: int, ldo: int) -> torch.Tensor:
    gh, gw = img.h // p, img.w // p
    if DISPATCH == "torch":
        return _tops().patchify(img.view(), p, ldo)
    rows = torch.empty((img.n * gh * gw, ldo), device=img.device, dtype=torch.float32)
    L.check(L.load().prv2_patchify(img.ptr, img.n, gh, gw, p, img.ld, rows.data_ptr(), ldo, _stream()), "patchify")
    return rows


def assemble_tokens(emb: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, b: int, np_: int, dim: int) -> torch.Tensor:
    if DISPATCH == "torch":
        return _tops().assemble_tokens(emb, cls, pos, b, np_, dim)
    tok = torch.empty((b, np_ + 1, dim), device=emb.device, dtype=torch.float32)
    L.check(L.load().prv2_assemble_tokens(emb.data_ptr(), cls.data_ptr(), pos.data_ptr(), b, np_, dim, tok.data_ptr(),
                                          _stream()), "assemble_tokens")
    return tok


class AttnBiasImage:
    """a [heads, ntok, ld] score bias re-ordered once for the bf16x3 attention kernel (include/prv2.h::prv2_pack_attention_bias)"""

    def __init__(self, image: torch.Tensor, heads: int, ntok: int):
        self.image, self.heads, self.ntok = image, heads, ntok


ATT_BIAS_IMAGE = os.environ.get("PRV2_ATT_BIAS_IMAGE", "1") != "0"  # A/B and test switch


def pack_attention_bias(bias: torch.Tensor, ntok: int):
    """bias rows [heads, ntok, ld] -> AttnBiasImage (bf16 modes; same values, the kernel's per-tile log2 e product pre-formed)"""
    _require_dev(bias)
    assert bias.is_contiguous() and bias.dim() == 3 and bias.shape[1] == ntok
    heads = bias.shape[0]
    if DISPATCH == "torch":
        return AttnBiasImage(_tops().pack_attention_bias(bias, ntok), heads, ntok)
    lib = L.load()
    img = torch.empty(lib.prv2_attention_bias_image_bytes(heads, ntok) // 4, device=bias.device, dtype=torch.float32)
    L.check(lib.prv2_pack_attention_bias(bias.data_ptr(), heads, ntok, bias.shape[2], img.data_ptr(), _stream()), "pack_attention_bias")
    return AttnBiasImage(img, heads, ntok)


def attention(qkv: torch.Tensor, b: int, ntok: int, heads: int, prec: int = PREC_F32, bias=None,
              out_ss: bool = False) -> torch.Tensor:
    """``bias``: optional [heads, ntok, ld >= roundup(ntok, 64)] additive score bias shared by the batch (BEiT), or its
    ``AttnBiasImage`` (bf16 modes); ``out_ss``: write the output split-swizzled (the operand format of gemm_ss; bf16x3 only)"""
    out = torch.empty((b * ntok, heads * 64), device=qkv.device, dtype=torch.float32)
    lib = L.load()
    nbytes = lib.prv2_attention_workspace_bytes(b, ntok, heads, prec)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device) if nbytes else None
    image = isinstance(bias, AttnBiasImage)
    if image:
        assert prec != PREC_F32 and (bias.heads, bias.ntok) == (heads, ntok)
        bias_t, ld_bias = bias.image, -1
    elif bias is not None:
        _require_dev(bias)
        assert bias.is_contiguous() and bias.shape[0] == heads and bias.shape[1] == ntok
        bias_t, ld_bias = bias, bias.shape[2]
    else:
        bias_t, ld_bias = None, 0
    if DISPATCH == "torch":
        res = []
        PROFILER.launch("attention_f32_kernel" if prec == PREC_F32 else "attention_bf16x3_kernel", 4.0 * b * heads * ntok * ntok * 64,
                        lambda: res.append(_tops().attention_ss(qkv, b, ntok, heads, bias_t, image) if out_ss else
                                           _tops().attention_fwd(qkv, b, ntok, heads, prec, bias_t, image)))
        return res[0]
    if out_ss:
        assert prec == L.PREC_BF16X3
        PROFILER.launch("attention_bf16x3_kernel", 4.0 * b * heads * ntok * ntok * 64,
                        lambda: L.check(lib.prv2_attention_ss(qkv.data_ptr(), b, ntok, heads, 64, _ptr(bias_t), ld_bias, out.data_ptr(), _ptr(ws),
                                                              nbytes, _stream()), "attention_ss"))
        return out
    PROFILER.launch("attention_f32_kernel" if prec == PREC_F32 else "attention_bf16x3_kernel",
                    4.0 * b * heads * ntok * ntok * 64,
                    lambda: L.check(lib.prv2_attention_bias(qkv.data_ptr(), b, ntok, heads, 64, _ptr(bias_t), ld_bias, out.data_ptr(), prec,
                                                            _ptr(ws), nbytes, _stream()), "attention"))
    return out


def crop_resize(img_chw: torch.Tensor, tiles: torch.Tensor, ch: int, cw: int, oh: int, ow: int,
                mean: Optional[Sequence[float]], std: Optional[Sequence[float]], out: Feat):
    """img_chw [3,H,W] device; tiles int32 [k,2] device; writes channels 0..2 of ``out`` [k,oh,ow,*]."""
    _require_dev(img_chw)
    assert tiles.dtype == torch.int32 and tiles.is_cuda and img_chw.is_contiguous()
    k = tiles.shape[0]
    if DISPATCH == "torch":
        _tops().crop_resize_bilinear(img_chw, tiles.contiguous(), ch, cw, oh, ow, list(mean) if mean is not None else None,
                                     list(std) if std is not None else None, out.slice(0, 3).view())
        return
    m = (C.c_float * 3)(*mean) if mean is not None else None
    s = (C.c_float * 3)(*std) if std is not None else None
    L.check(L.load().prv2_crop_resize(img_chw.data_ptr(), img_chw.shape[1], img_chw.shape[2], tiles.data_ptr(), k, ch,
                                      cw, oh, ow, m, s, out.ptr, out.ld, _stream()), "crop_resize")


def bicubic_resize(img_hwc: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """[h, w, 3] uint8 (-> /255) or fp32 RGB image on the device -> [3, oh, ow] fp32 (bicubic, align_corners=True);
    general_dataset.py:55-60."""
    if not img_hwc.is_cuda or img_hwc.dtype not in (torch.uint8, torch.float32):
        raise ValueError("bicubic_resize needs a uint8 or float32 image on the GPU (no CPU fallback exists)")
    assert img_hwc.dim() == 3 and img_hwc.shape[2] == 3
    img_hwc = img_hwc.contiguous()
    if DISPATCH == "torch":
        return _tops().bicubic_resize(img_hwc, oh, ow)
    out = torch.empty((3, oh, ow), device=img_hwc.device, dtype=torch.float32)
    L.check(L.load().prv2_bicubic_resize(img_hwc.data_ptr(), int(img_hwc.dtype == torch.uint8), img_hwc.shape[0], img_hwc.shape[1],
                                         out.data_ptr(), oh, ow, _stream()), "bicubic_resize")
    return out


def roi_align(feat: Feat, boxes: torch.Tensor, spatial_scale: float, oh: int, ow: int, out: Optional[Feat] = None) -> Feat:
    assert feat.n == 1 and boxes.dtype == torch.float32 and boxes.is_cuda and boxes.shape[1] == 4
    k = boxes.shape[0]
    if out is None:
        out = Feat.alloc(k, oh, ow, feat.c, feat.device)
    if DISPATCH == "torch":
        PROFILER.launch_aux("roi_align", 4.0 * feat.c * (feat.h * feat.w + k * oh * ow),
                            lambda: _tops().roi_align(feat.view(), boxes.contiguous(), float(spatial_scale), oh, ow, out.raw(), out.x2),
                            f"{feat.c}ch {feat.h}x{feat.w}->{k}x{oh}x{ow}{' x2' if out.x2 else ''}")
        return out
    fn = L.load().prv2_roi_align_x2 if out.x2 else L.load().prv2_roi_align  # (x2: the pre-split format of the gate kernel's input)
    PROFILER.launch_aux("roi_align", 4.0 * feat.c * (feat.h * feat.w + k * oh * ow),
                        lambda: L.check(fn(feat.ptr, feat.h, feat.w, feat.c, feat.ld, boxes.data_ptr(), k,
                                           spatial_scale, oh, ow, out.ptr, out.ld, _stream()), "roi_align"),
                        f"{feat.c}ch {feat.h}x{feat.w}->{k}x{oh}x{ow}{' x2' if out.x2 else ''}")
    return out


def upsample_bilinear(x: Feat, oh: int, ow: int, out: Optional[Feat] = None) -> Feat:
    if out is None:
        out = Feat.alloc(x.n, oh, ow, x.c, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, oh, ow, x.c)
    if DISPATCH == "torch":
        PROFILER.launch_aux("upsample_bilinear", 4.0 * x.n * x.c * (x.h * x.w + oh * ow),
                            lambda: _tops().upsample_bilinear_ac(x.view(), oh, ow, out.view()), f"{x.c}ch {x.n}x{x.h}x{x.w}->{oh}x{ow}")
        return out
    PROFILER.launch_aux("upsample_bilinear", 4.0 * x.n * x.c * (x.h * x.w + oh * ow),
                        lambda: L.check(L.load().prv2_upsample_bilinear(x.ptr, x.n, x.h, x.w, x.c, x.ld, oh, ow, out.ptr,
                                                                         out.ld, _stream()), "upsample_bilinear"),
                        f"{x.c}ch {x.n}x{x.h}x{x.w}->{oh}x{ow}")
    return out


def _pre_desc(x: Feat, cw: ConvW, out_ld: int, act: int, res_ld: int, ln_eps: float):
    return L.ConvDesc(n=x.n, h=x.h, w=x.w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad, ldx=x.ld, ldy=out_ld,
                      x_bstride=0, y_bstride=0, relu_in=0, act=act, convt_k=cw.convt_k, ld_mul=0, ld_res=res_ld, ld_res2=0, prec=cw.prec,
                      force_generic=0, ln_eps=ln_eps, part=0, same_pad=int(cw.same_pad), fmt=0)


def conv2d_pre_supported(h: int, w: int, cw: ConvW, ln: bool) -> bool:
    """can ``conv2d_pre`` add a pre-epilogue addend to this 3x3 conv on h x w images (a property of the layer)?"""
    if ln and cw.cout > 128 and cw.cout != 256:
        return False
    d = L.ConvDesc(n=1, h=h, w=w, cin=cw.cin, cout=cw.cout, kh=cw.kh, kw=cw.kw, stride=cw.stride, pad=cw.pad, ldx=roundup(cw.cin, 4),
                   ldy=roundup(cw.cout, 4), x_bstride=0, y_bstride=0, relu_in=0, act=ACT_NONE, convt_k=cw.convt_k, ld_mul=0, ld_res=0, ld_res2=0,
                   prec=cw.prec, force_generic=0, ln_eps=1e-6, part=0, same_pad=int(cw.same_pad), fmt=0)
    return bool(L.load().prv2_conv2d_pre_supported(C.byref(d)))


def conv2d_pre(x: Feat, cw: ConvW, pre: Feat, out: Optional[Feat] = None, *, act: int = ACT_NONE, res: Optional[Feat] = None, ln=None,
               ln_eps: float = 1e-6, pre_cin: int = 0) -> Feat:
    """y = act([LN](conv3x3(x) + pre + bias)) (+ res) (include/prv2.h::prv2_conv2d_pre): ``pre`` [n, h, w, cout] is the coarse half of the
    reference's conv over cat([coarse_roi, x]) from ``CoarseTaps.gather``, standing for ``pre_cin`` further input channels"""
    assert x.c == cw.cin and (pre.n, pre.h, pre.w, pre.c) == (x.n, x.h, x.w, cw.cout) and not pre.x2
    if out is None:
        out = Feat.alloc(x.n, x.h, x.w, cw.cout, x.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, cw.cout) and not out.x2
    d = _pre_desc(x, cw, out.ld, act, res.ld if res is not None else 0, ln_eps)
    flops = 2.0 * x.n * x.h * x.w * cw.cout * cw.cin * 9
    def call():
        if DISPATCH == "torch":
            _tops().conv3x3_pre(x.view(), cw.w, cw.bias, pre.view(), cw.cout, act, ln[0] if ln is not None else None, ln[1] if ln is not None else None,
                                res.view() if res is not None else None, cw.prec, ln_eps, out.view())
            return
        L.check(L.load().prv2_conv2d_pre(C.byref(d), x.ptr, cw.w.data_ptr(), _ptr(cw.bias), pre.ptr, pre.ld, _ptr(ln[0]) if ln is not None else None,
                                         _ptr(ln[1]) if ln is not None else None, _ptr(res), out.ptr, _stream()), "conv2d_pre")

    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), flops, call,
                    shape=f"{cw.cin}(+{pre_cin} coarse)->{cw.cout} k3s1 {x.n}x{x.h}x{x.w}", algo=flops + 2.0 * x.n * x.h * x.w * cw.cout * pre_cin * 9)
    return out


CHAIN32 = os.environ.get("PRV2_CHAIN32", "1") != "0"  # A/B and test switch: the fused 32-channel full-resolution chains (csrc/chain32.hip)


def pack_chain32(weight: torch.Tensor, kind: int, device=None) -> torch.Tensor:
    """fragment image of a 32-output-channel conv for the ``chain32_*`` kernels (include/prv2.h::prv2_pack_chain32_weight): kind 0 = first
    conv of a chain (3x3, natural K), 1 = second conv / 1x1 (K in accumulator order), 2 = the [p1 | p2] tail of a 3x3 over 34 channels"""
    w = weight.detach().to(device=device or weight.device, dtype=torch.float32).contiguous()
    assert w.shape[0] == 32 and w.dim() in (2, 4)
    taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
    if DISPATCH == "torch":
        return _tops().pack_chain32_weight(w, kind)
    lib = L.load()
    packed = torch.empty(lib.prv2_chain32_weight_bytes(kind, taps) // 4, device=w.device, dtype=torch.float32)
    L.check(lib.prv2_pack_chain32_weight(w.data_ptr(), w.shape[1], taps, kind, packed.data_ptr(), _stream()), "pack_chain32_weight")
    return packed


def chain32_consts(device, **rows) -> torch.Tensor:
    """the [9][32] constant table of the ``chain32_*`` kernels: b1, ln1w, ln1b, b2, bg, bo, w3, ln2w, ln2b (missing rows zero)"""
    names = ("b1", "ln1w", "ln1b", "b2", "bg", "bo", "w3", "ln2w", "ln2b")
    assert set(rows) <= set(names)
    t = torch.zeros((9, 32), dtype=torch.float32, device=device)
    for i, n in enumerate(names):
        if rows.get(n) is not None:
            t[i] = rows[n].detach().to(device=device, dtype=torch.float32).reshape(32)
    return t


def _chain32_desc(x: Feat, w1, w2, wg, wo, consts, pre, p1, p2, y: Feat, depth, b3, ln_eps):
    assert x.c == 32 and y.c == 32 and (x.n, x.h, x.w) == (y.n, y.h, y.w) and not x.x2 and not y.x2
    assert pre is None or ((pre.n, pre.h, pre.w, pre.c) == (x.n, x.h, x.w, 32) and not pre.x2)
    return L.Chain32Desc(x=x.ptr, w1=w1.data_ptr(), w2=w2.data_ptr(), wg=wg.data_ptr(), wo=_ptr(wo), consts=consts.data_ptr(), pre=_ptr(pre),
                         p1=_ptr(p1), p2=_ptr(p2), y=y.ptr, depth=_ptr(depth), x_bstride=0, y_bstride=0, n=x.n, h=x.h, w=x.w, ldx=x.ld, ldy=y.ld,
                         ld_pre=pre.ld if pre is not None else 0, b3=b3, ln_eps=ln_eps)


def chain32_c2f(x: Feat, cw: dict, pre: Optional[Feat], out: Optional[Feat] = None, depth: Optional[torch.Tensor] = None, ln_eps: float = 1e-6,
                pre_cin: int = 0):
    """C2FModule's full-resolution tail in one kernel (include/prv2.h::prv2_chain32_c2f): GateresConfUnit2 of ``output_conv2_fusion`` (conv +
    skip, fusion_conv with the coarse half ``pre``, gate), ``out_conv`` and ``output_conv3``.  cw: dict(w1, w2, wg, wo, consts, b3) of
    ``pack_chain32`` / ``chain32_consts`` images.  Returns (last feature [n, h, w, 32], depth [n, 1, h, w])."""
    if out is None:
        out = Feat(torch.empty((x.n, x.h, x.w, 32), device=x.device, dtype=torch.float32))
    if depth is None:
        depth = torch.empty((x.n, 1, x.h, x.w), device=x.device, dtype=torch.float32)
    assert depth.is_contiguous() and depth.numel() == x.n * x.h * x.w

    def call():
        if DISPATCH == "torch":
            _tops().chain32_c2f(x.view(), cw["w1"], cw["w2"], cw["wg"], cw["wo"], cw["consts"], cw["b3"], pre.view() if pre is not None else None, ln_eps,
                                out.view(), depth)
            return
        d = _chain32_desc(x, cw["w1"], cw["w2"], cw["wg"], cw["wo"], cw["consts"], pre, None, None, out, depth, cw["b3"], ln_eps)
        L.check(L.load().prv2_chain32_c2f(C.byref(d), _stream()), "chain32_c2f")

    px = float(x.n * x.h * x.w)
    flops = 2.0 * px * (2 * 9 * 32 * 32 + 2 * 32 * 32 + 32)
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), flops, call, shape=f"32->32->32(+{pre_cin} coarse)->gate->32->1 k3s1 {x.n}x{x.h}x{x.w}",
                    algo=flops + 2.0 * px * 9 * 32 * pre_cin)
    return out, depth


def chain32_enc(x: Feat, cw: dict, pre: Feat, p1: torch.Tensor, p2: torch.Tensor, out: Optional[Feat] = None, ln_eps: float = 1e-6, pre_cin: int = 0) -> Feat:
    """``fusion_layers_1[0]`` + ``fusion_layers_2[0]`` in one kernel (include/prv2.h::prv2_chain32_enc).  cw: dict(w1, w2, wt, consts);
    p1 / p2: dense depth maps at x's size."""
    if out is None:
        out = Feat(torch.empty((x.n, x.h, x.w, 32), device=x.device, dtype=torch.float32))
    _require_dev(p1, p2)
    assert p1.is_contiguous() and p2.is_contiguous() and p1.numel() == p2.numel() == x.n * x.h * x.w

    def call():
        if DISPATCH == "torch":
            _tops().chain32_enc(x.view(), cw["w1"], cw["w2"], cw["wt"], cw["consts"], pre.view(), p1, p2, ln_eps, out.view())
            return
        d = _chain32_desc(x, cw["w1"], cw["w2"], cw["wt"], None, cw["consts"], pre, p1, p2, out, None, 0.0, ln_eps)
        L.check(L.load().prv2_chain32_enc(C.byref(d), _stream()), "chain32_enc")

    px = float(x.n * x.h * x.w)
    flops = 2.0 * px * (9 * 32 * 32 + 9 * 34 * 32)
    PROFILER.launch(lambda: L.load().prv2_last_kernel().decode(), flops, call, shape=f"32(+{pre_cin} coarse)->32->34->32 k3s1 {x.n}x{x.h}x{x.w}",
                    algo=flops + 2.0 * px * 9 * 32 * pre_cin)
    return out


COARSE_TAPS = os.environ.get("PRV2_COARSE_TAPS", "1") != "0"  # A/B and test switch: coarse half of the cat([fine, coarse_roi]) convs once per frame


class CoarseTaps:
    """The coarse half of one ``cat([fine, coarse_roi])`` 3x3 conv, tabulated once per frame (include/prv2.h::prv2_coarse_tap_knots):
    ``g`` [1, H, W, 9 * cout] = the level's map through the conv's coarse weights, tap-major (a 1x1 GEMM at coarse resolution),
    ``v`` [1, 3H, 3W, cout] = the unmasked tap sum on the knot grid, ``kb`` = (tile height / frame height, tile width / frame width)."""

    def __init__(self, g: Feat, cout: int, kb):
        assert g.n == 1 and g.c == 9 * cout and 0 < kb[0] <= 0.5 and 0 < kb[1] <= 0.5
        self.g, self.cout, self.kb = g, cout, (float(kb[0]), float(kb[1]))
        if DISPATCH == "torch":
            box = []
            PROFILER.launch_aux("coarse_tap_knots", 4.0 * g.h * g.w * 18 * cout, lambda: box.append(_tops().coarse_tap_knots(g.view(), cout, self.kb[0], self.kb[1])),
                                f"{cout}ch {g.h}x{g.w}")
            self.v = Feat(box[0])
            return
        self.v = Feat(torch.empty((1, 3 * g.h, 3 * g.w, cout), device=g.device, dtype=torch.float32))
        PROFILER.launch_aux("coarse_tap_knots", 4.0 * g.h * g.w * 18 * cout,
                            lambda: L.check(L.load().prv2_coarse_tap_knots(g.ptr, g.h, g.w, cout, g.ld, self.kb[0], self.kb[1], self.v.ptr, self.v.ld,
                                                                           _stream()), "coarse_tap_knots"), f"{cout}ch {g.h}x{g.w}")

    def gather(self, boxes: torch.Tensor, spatial_scale: float, oh: int, ow: int, out: Optional[Feat] = None) -> Feat:
        """the conv's coarse half for the tiles ``boxes`` (as roi_align takes them): [k, oh, ow, cout], zero padding at the tile border
        included (prv2_coarse_tap_gather)"""
        assert boxes.dtype == torch.float32 and boxes.is_cuda and boxes.shape[1] == 4
        k = boxes.shape[0]
        # The knot-grid algebra holds when consecutive output pixels are exactly ``kb`` coarse pixels apart (ROI bin == knot spacing): every
        # box must span kb * (g.h, g.w) * (oh, ow) / spatial_scale frame pixels.  The boxes live on the device (no host sync on the frame
        # path): PRV2_CHECK_TAPS=1 verifies them here; the library checks what it can see (knot spacing in (0, 1/2]), fusion.py compares the
        # ROI's output size with its consumer's map before taking this path.
        if os.environ.get("PRV2_CHECK_TAPS"):
            b = boxes.detach().cpu()
            bin_w = (b[:, 2] - b[:, 0]) * spatial_scale / ow
            bin_h = (b[:, 3] - b[:, 1]) * spatial_scale / oh
            assert float((bin_h - self.kb[0]).abs().max()) < 1e-4 and float((bin_w - self.kb[1]).abs().max()) < 1e-4, \
                ("CoarseTaps.gather: ROI bin size differs from the knot spacing the table was built for", float(bin_h[0]), float(bin_w[0]), self.kb)
        if out is None:
            out = Feat(torch.empty((k, oh, ow, self.cout), device=self.g.device, dtype=torch.float32))
        assert (out.n, out.h, out.w, out.c) == (k, oh, ow, self.cout) and not out.x2
        g, v = self.g, self.v
        if DISPATCH == "torch":
            PROFILER.launch_aux("coarse_tap_gather", 4.0 * self.cout * (9 * g.h * g.w + k * oh * ow),
                                lambda: _tops().coarse_tap_gather(v.view(), g.view(), self.kb[0], self.kb[1], boxes.contiguous(), float(spatial_scale), oh, ow, out.view()),
                                f"{self.cout}ch {g.h}x{g.w}->{k}x{oh}x{ow}")
            return out
        PROFILER.launch_aux("coarse_tap_gather", 4.0 * self.cout * (9 * g.h * g.w + k * oh * ow),
                            lambda: L.check(L.load().prv2_coarse_tap_gather(v.ptr, g.ptr, g.h, g.w, self.cout, v.ld, g.ld, self.kb[0], self.kb[1],
                                                                            boxes.data_ptr(), k, spatial_scale, oh, ow, out.ptr, out.ld, _stream()),
                                            "coarse_tap_gather"), f"{self.cout}ch {g.h}x{g.w}->{k}x{oh}x{ow}")
        return out


def coarse_tap_weight(w_coarse: torch.Tensor) -> torch.Tensor:
    """[cout, cin, 3, 3] coarse-half weights of a conv -> the 1x1 GEMM weights [9 * cout, cin] whose output is tap-major
    (row tap * cout + co = w[co, :, ky, kx]): G of ``CoarseTaps``"""
    co, ci = w_coarse.shape[:2]
    return w_coarse.permute(2, 3, 0, 1).reshape(9 * co, ci).contiguous()


DIRECT_PLACEMENT = os.environ.get("PRV2_DIRECT_PLACEMENT", "1") != "0"  # A/B and test switch: ROI levels / depth pairs written by their producers


class RoiSource:
    """A level of the coarse pyramid as seen by one batch of tiles: ``roi_align(feat.repeat(K), boxes, (h, w), scale)`` that has
    not been materialised.  Its consumers are concat buffers (the [coarse | fine] inputs of the fusion convs): ``write(dst)``
    gathers straight into a destination slice -- reading the small, L2-resident coarse map again is cheaper than writing the
    K-tile ROI once and copying it into each consumer (1 + 2 x (read + write) of K x c x h x w floats -> 2 writes)."""

    def __init__(self, feat: Feat, boxes: torch.Tensor, spatial_scale: float, oh: int, ow: int):
        self.feat, self.boxes, self.scale = feat, boxes, spatial_scale
        self.n, self.h, self.w, self.c = boxes.shape[0], oh, ow, feat.c
        self._mat: Optional[Feat] = None

    @property
    def device(self):
        return self.feat.device

    def write(self, dst: Feat):
        assert (dst.n, dst.h, dst.w, dst.c) == (self.n, self.h, self.w, self.c)
        if not DIRECT_PLACEMENT:
            return upsample_bilinear(self.materialize(), dst.h, dst.w, out=dst)
        roi_align(self.feat, self.boxes, self.scale, self.h, self.w, out=dst)

    def materialize(self) -> Feat:
        if self._mat is None:
            self._mat = roi_align(self.feat, self.boxes, self.scale, self.h, self.w)
        return self._mat


def conv_border_bias(y: Feat, tap_bias: torch.Tensor):
    """y -= the folded bias of the 3x3 taps that the zero padding hides at the image border (include/prv2.h::prv2_conv_border_bias)"""
    assert tap_bias.shape == (9, y.c) and tap_bias.is_contiguous()
    if DISPATCH == "torch":
        return PROFILER.launch_aux("conv_border_bias", 8.0 * y.n * 2 * (y.h + y.w) * y.c, lambda: _tops().conv_border_bias_(y.view(), tap_bias), f"{y.c}ch {y.n}x{y.h}x{y.w}")
    PROFILER.launch_aux("conv_border_bias", 8.0 * y.n * 2 * (y.h + y.w) * y.c,
                        lambda: L.check(L.load().prv2_conv_border_bias(y.ptr, y.n, y.h, y.w, y.c, y.ld, tap_bias.data_ptr(), _stream()),
                                        "conv_border_bias"), f"{y.c}ch {y.n}x{y.h}x{y.w}")


def depth_pair_fill(p1: Feat, p2: Feat, buf: Feat, c0: int):
    """channels c0, c0 + 1 of ``buf`` <- (p1, p2) resized to the buffer's size, channels c0 + 2, c0 + 3 (the pad) <- 0"""
    assert p1.c == 1 and p2.c == 1 and p1.ld == 1 and p2.ld == 1 and (p1.n, p1.h, p1.w) == (p2.n, p2.h, p2.w) == (buf.n, p1.h, p1.w)
    assert buf.ld == buf.c0 + c0 + 4 and (buf.c0 + c0) % 4 == 0, (buf.ld, buf.c0, c0)
    if DISPATCH == "torch":
        return PROFILER.launch_aux("depth_pair_fill", 16.0 * buf.n * buf.h * buf.w,
                                   lambda: _tops().depth_pair_fill(p1.buf.view(p1.n, p1.h, p1.w), p2.buf.view(p2.n, p2.h, p2.w), buf.buf[..., buf.c0 + c0:buf.c0 + c0 + 4]),
                                   f"{buf.n}x{p1.h}x{p1.w}->{buf.h}x{buf.w}")
    PROFILER.launch_aux("depth_pair_fill", 16.0 * buf.n * buf.h * buf.w,
                        lambda: L.check(L.load().prv2_depth_pair_fill(p1.ptr, p2.ptr, p1.n, p1.h, p1.w, buf.h, buf.w, buf.ptr + 4 * c0, buf.ld,
                                                                      _stream()), "depth_pair_fill"), f"{buf.n}x{p1.h}x{p1.w}->{buf.h}x{buf.w}")


def blend_paste(avg, cnt, pred, mask, tiles, th, tw):
    if DISPATCH == "torch":
        return _tops().blend_init(avg, cnt, pred.contiguous(), mask, tiles.contiguous(), th, tw)
    L.check(L.load().prv2_blend_paste(avg.data_ptr(), cnt.data_ptr(), avg.shape[0], avg.shape[1], pred.data_ptr(),
                                      pred.shape[-2], pred.shape[-1], mask.data_ptr(), tiles.data_ptr(), tiles.shape[0],
                                      th, tw, _stream()), "blend_paste")


def blend_update(avg, cnt, pred, mask, tiles, th, tw):
    if DISPATCH == "torch":
        return _tops().blend_update(avg, cnt, pred.contiguous(), mask, tiles.contiguous(), th, tw)
    L.check(L.load().prv2_blend_update(avg.data_ptr(), cnt.data_ptr(), avg.shape[0], avg.shape[1], pred.data_ptr(),
                                       pred.shape[-2], pred.shape[-1], mask.data_ptr(), tiles.data_ptr(), tiles.shape[0],
                                       th, tw, _stream()), "blend_update")


def blend_resize(avg, cnt, oh, ow):
    if DISPATCH == "torch":
        return _tops().blend_resize(avg, cnt, oh, ow)
    a = torch.empty((oh, ow), device=avg.device, dtype=torch.float32)
    c = torch.empty((oh, ow), device=avg.device, dtype=torch.float32)
    L.check(L.load().prv2_blend_resize(avg.data_ptr(), cnt.data_ptr(), avg.shape[0], avg.shape[1], a.data_ptr(),
                                       c.data_ptr(), oh, ow, _stream()), "blend_resize")
    return a, c


def add(a: Feat, b: Feat, out: Optional[Feat] = None) -> Feat:
    assert (a.n, a.h, a.w, a.c) == (b.n, b.h, b.w, b.c)
    if out is None:
        out = Feat.alloc(a.n, a.h, a.w, a.c, a.device)
    if DISPATCH == "torch":
        _tops().add_nhwc(a.view(), b.view(), out.view())
        return out
    L.check(L.load().prv2_add(a.ptr, a.ld, b.ptr, b.ld, a.n * a.h * a.w, a.c, out.ptr, out.ld, _stream()), "add")
    return out


def zoe_attractor(attr: Feat, bins: Feat, alpha: float = 300.0) -> Feat:
    assert (attr.n, attr.h, attr.w) == (bins.n, bins.h, bins.w)
    if DISPATCH == "torch":
        y = _tops().zoe_attractor(attr.view(), bins.view(), alpha)
        return _feat_of(y, bins.c)
    out = Feat.alloc(bins.n, bins.h, bins.w, bins.c, bins.device)
    L.check(L.load().prv2_zoe_attractor(attr.ptr, attr.ld, attr.c, bins.ptr, bins.ld, bins.c, alpha,
                                        bins.n * bins.h * bins.w, out.ptr, out.ld, _stream()), "zoe_attractor")
    return out


def zoe_logbinom_depth(pt: Feat, centers: Feat, min_temp: float, max_temp: float) -> torch.Tensor:
    assert pt.c == 4 and (pt.n, pt.h, pt.w) == (centers.n, centers.h, centers.w)
    if DISPATCH == "torch":
        return _tops().zoe_bins_head(pt.view(), centers.view(), min_temp, max_temp)
    depth = torch.empty((pt.n, 1, pt.h, pt.w), device=pt.device, dtype=torch.float32)
    L.check(L.load().prv2_zoe_logbinom_depth(pt.ptr, pt.ld, centers.ptr, centers.ld, centers.c, min_temp, max_temp,
                                             pt.n * pt.h * pt.w, depth.data_ptr(), _stream()), "zoe_logbinom_depth")
    return depth
