"""A hazard hipcc's recognizer cannot see (tools/isa_hazard_scan.py): inline-asm VMEM instructions reading an SGPR that a VALU instruction -- the
restore of a spilled scalar by v_readlane -- wrote fewer than five wait states earlier.  Met once (profiles/r04_experiments.txt #10 d): the bf16
instantiation of upconv3x3_kernel fetched weight pieces from a stale address.  The scan compiles the sources that issue VMEM from inline asm with
scalar operands to gfx950 ISA (no GPU needed) and must stay clean."""
import importlib.util
import os

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_hazard_scan", os.path.join(HERE, "tools", "isa_hazard_scan.py"))
scan = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scan)

ASM = """
_ZN4prv21kEv:
\tv_readlane_b32 s4, v255, 3
\tv_readlane_b32 s5, v255, 4
\ts_mov_b32 m0, s22
\t;;#ASMSTART
\ts_nop {nop}
\tglobal_load_lds_dwordx4 v22, s[4:5]
\t;;#ASMEND
\tv_readfirstlane_b32 s8, v1
\tbuffer_load_dwordx4 v[0:3], v4, s[8:11], 0 offen
"""


def test_scanner_sees_the_hazard_and_the_wait_states_that_cure_it():
    hz = scan.scan_asm(ASM.format(nop=0))
    assert [(k, ws) for k, _, _, ws in hz] == [("_ZN4prv21kEv", 3), ("_ZN4prv21kEv", 2)]  # s4 and s5; the compiler-emitted buffer_load is its own business
    assert scan.scan_asm(ASM.format(nop=4)) == []


def test_no_inline_asm_vmem_reads_a_freshly_restored_sgpr():
    files = scan.sources_with_inline_vmem()
    assert any(f.endswith("upconv.hip") for f in files) and any(f.endswith("conv3x3_m16.hip") for f in files)
    assert scan.main(files) == 0


def test_no_packed_fp32_math_in_the_gather_family():
    """the only mitigation of tap_gather_kernel's intermittent border-pixel miscompute is "no v_pk_*_f32 in that code" -- it lives in
    the SOURCES now (common.h PRV2_NO_PACKED_FP32_BEGIN, a device-pass function attribute), and this asserts the ISA: zero packed fp32
    instructions in every kernel of coarse_taps / gather / blend / pointwise, and -- positive control -- thousands in coarse_taps.hip when the
    attribute is disabled (-DPRV2_TAPS_PK, what ``make TAPS_PK=1`` builds for the negative-control GPU test)."""
    for fn in scan.NO_PACKED_FP32_SOURCES:
        counts = scan.packed_fp32_by_kernel(os.path.join(scan.CSRC, fn))
        assert counts and all(v == 0 for v in counts.values()), (fn, {k: v for k, v in counts.items() if v})
    taps = scan.packed_fp32_by_kernel(os.path.join(scan.CSRC, "coarse_taps.hip"))
    assert any("tap_gather_kernel" in k for k in taps) and any("tap_knots_kernel" in k for k in taps), list(taps)
    pk = scan.packed_fp32_by_kernel(os.path.join(scan.CSRC, "coarse_taps.hip"), defines=("-DPRV2_TAPS_PK",))
    assert sum(v for k, v in pk.items() if "tap_gather_kernel" in k) > 100, pk
