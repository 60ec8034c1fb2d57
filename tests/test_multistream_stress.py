"""Multi-stream determinism of the benchmarked frame: the intermittent miscompute of hipcc's packed-fp32 code for
tap_gather_kernel (csrc/coarse_taps.hip BUILD NOTE) only ever showed inside frames with two tile streams and a third stream busy.
  (a) negative control: the same frames on a scratch build WITH the packed code (child process) -- the detector must see it fail;
  (b) a repeated-frame stress loop on the shipped build: bit-identical frames;
  (c) every conv / gather kernel family beside the per-frame tap preparation on another stream: bit-identical to running alone."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(script, *args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", script), *map(str, args)], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_shipped_build_three_stream_frames_are_bit_stable():
    """(b) 12 frames alternating between two images, each with the other's coarse forward prefetched beside its tiles (41-tile batches, 3 streams)"""
    out = _child("taps_pk_negative_control.py", 12, "shipped")
    m = re.search(r"RESULT library=shipped frames=12 mismatching=(\d+)", out)
    assert m and int(m.group(1)) == 0, out[-2000:]


def test_packed_fp32_build_is_caught_by_the_same_loop():
    """(a) the detector is not blind: with coarse_taps.hip's packed fp32 math back on, the same loop reports mismatching frames.  The fault is
    intermittent and box dependent: not reproducing within 20 frames is reported as a skip, never as a pass."""
    out = _child("taps_pk_negative_control.py", 20, "pk")
    m = re.search(r"RESULT library=pk frames=20 mismatching=(\d+)", out)
    assert m, out[-2000:]
    if int(m.group(1)) == 0:
        pytest.skip("the packed-fp32 build did not miscompute within 20 frames on this box (intermittent: profiles/r04_experiments.txt #1)")


def test_kernel_families_beside_the_tap_preparation():
    """(c) tools/probes/victims_under_prep.py: every conv / depthwise / upsample / single-output-channel kernel family gives the same bits with the
    per-frame tap preparation (a 4608-column GEMM + the knot-table kernel) running beside it on another stream"""
    out = _child("victims_under_prep.py")
    rows = re.findall(r"^(\d+)/5\s+(.*)$", out, flags=re.M)
    assert len(rows) >= 15, out[-2000:]
    assert all(int(b) == 0 for b, _ in rows), [r for r in rows if int(r[0])]
