"""Static scan of the gfx950 ISA of the HIP sources for a hazard hipcc cannot see: a VMEM instruction INSIDE an inline-asm statement that reads an
SGPR written by a VALU instruction (v_readlane / v_readfirstlane -- e.g. the restore of a spilled scalar) fewer than 5 wait states earlier.  The
hazard recognizer inserts the wait states for instructions it emits itself, but does not look into inline asm (profiles/r04_experiments.txt #10 d:
the bf16 instantiation of upconv3x3_kernel fetched weight pieces from a stale address).  Also reports scratch use per kernel.

    python tools/isa_hazard_scan.py [file.hip ...]      -> exit code 1 if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(HERE, "patchrefinerv2_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-I" + CSRC]
VMEM = re.compile(r"^(buffer_|global_|scratch_|flat_)")


def scan_asm(text):
    """[(kernel, line of the SGPR write, line of the VMEM read, wait states in between)] for VMEM reads inside inline asm"""
    out, kernel, in_asm = [], None, False
    ins = []  # (line no, text, inside inline asm, kernel)
    for n, raw in enumerate(text.split("\n"), 1):
        l = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)  # (function label, possibly with a trailing "; @name" comment)
        if m:
            kernel = m.group(1)
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not raw.startswith("\t") or not l or l[0] in ";." or l.endswith(":") or m:
            continue
        ins.append((n, l, in_asm, kernel))
    for i, (n, l, _, k) in enumerate(ins):
        m = re.match(r"v_read(?:first)?lane_b32 s(\d+),", l)
        if not m:
            continue
        s, ws = int(m.group(1)), 0
        for n2, l2, asm2, _ in ins[i + 1:i + 8]:
            if l2.startswith("s_nop"):
                ws += int(l2.split()[1]) + 1
            else:
                if asm2 and VMEM.match(l2):
                    regs = [(int(a), int(b)) for a, b in re.findall(r"s\[(\d+):(\d+)\]", l2)] + [(int(a), int(a)) for a in re.findall(r"\bs(\d+)\b", l2)]
                    if any(a <= s <= b for a, b in regs):
                        out.append((k, n, n2, ws))
                        break
                ws += 1
            if ws >= 5:
                break
    return out


def compile_asm(path):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        tmp = f.name
    try:
        r = subprocess.run([HIPCC] + FLAGS + [path, "-o", tmp], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{path}: {r.stderr[-2000:]}")
        return open(tmp).read()
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


NO_PACKED_FP32_SOURCES = ("coarse_taps.hip", "gather.hip", "blend.hip", "pointwise.hip")  # common.h PRV2_NO_PACKED_FP32_BEGIN


def packed_fp32_by_kernel(path, defines=()):
    """{kernel: number of v_pk_*_f32 instructions} of a source compiled as the Makefile compiles it (+ ``defines``).  The files above switch
    packed fp32 math off in the source (hipcc 7.2's v_pk_fma_f32 code for tap_gather_kernel miscomputed under multi-stream load,
    csrc/coarse_taps.hip BUILD NOTE): their count must stay zero whatever the build route."""
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        tmp = f.name
    try:
        r = subprocess.run([HIPCC] + FLAGS + list(defines) + [path, "-o", tmp], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{path}: {r.stderr[-2000:]}")
        text = open(tmp).read()
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    out, kernel = {}, None
    for raw in text.split("\n"):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel = m.group(1)
            out.setdefault(kernel, 0)
        elif kernel and re.match(r"^\s+v_pk_[a-z0-9]+_f32\b", raw):
            out[kernel] += 1
    return out


def sources_with_inline_vmem():
    """the .hip files whose inline asm issues VMEM instructions with scalar operands"""
    out = []
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith(".hip"):
            src = open(os.path.join(CSRC, fn)).read()
            if re.search(r'asm volatile\("[^;]*?(buffer_load|global_load|global_store)[^;]*?"s"\(', src, re.S):
                out.append(os.path.join(CSRC, fn))
    return out


def main(argv):
    files = argv or sources_with_inline_vmem()
    with ThreadPoolExecutor(max_workers=min(8, len(files) or 1)) as ex:
        asms = list(ex.map(compile_asm, files))
    bad = 0
    for path, text in zip(files, asms):
        hz = scan_asm(text)
        scratch = re.findall(r"; ScratchSize: (\d+)", text)
        print(f"{os.path.basename(path)}: {len(hz)} hazard(s); kernels with scratch: {sum(1 for s in scratch if int(s) > 0)} of {len(scratch)}")
        for k, n, n2, ws in hz:
            print(f"    {k}: SGPR written by VALU at line {n}, read by inline-asm VMEM at line {n2} after {ws} wait state(s)")
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
