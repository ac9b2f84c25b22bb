"""k_fill's compact layout has no LDS for the bitmap of SLOW bodies and derives the flags from the band ends as they stream by
(poreseq_amd/csrc/ps_slowmask.h, fill_body's chunk loop).  Host check of that logic against the up-front rule, over random masks
and whole random sweeps; the GPU tests run both layouts on bands that jump and resume (test_hip_parity.py)."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_streamed_slow_flags_equal_the_bitmap_rule():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "slowmask_check")
        subprocess.check_call(["g++", "-O2", os.path.join(HERE, "native", "slowmask_check.cpp"), "-o", exe])
        out = subprocess.check_output([exe, "400000"], timeout=300).decode()
    assert out.strip().endswith("mismatches=0"), out
