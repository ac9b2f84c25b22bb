"""The strip sweeps keep seven predicate bits per cell, packed four rows to a register and stored plane by plane (poreseq_amd/csrc/ps_codes.h,
ps_sweep_body.h: code_push / put_codes); the backtrace and the debug dump read them back with code_fetch.  Host check of the layout for every
strip height and lane count the kernels are built for, and of the decoders' truth tables; the GPU tests compare the decoded step codes of
whole DP matrices with the oracle (test_hip_parity.py, test_hip_regime.py)."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_code_fields_round_trip_for_every_strip_height():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "codes_check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, "native", "codes_check.cpp"), "-o", exe])
        out = subprocess.check_output([exe], timeout=120).decode()
    assert out.strip().endswith("mismatches=0"), out
