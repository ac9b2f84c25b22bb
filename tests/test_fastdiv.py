"""k_fill forms its three quotients per emission from tabulated reciprocals with Markstein's FMA sequence (ps_kernels.hip, mdiv).
The sequence is provably the correctly rounded quotient (y = RN(1/b), faithful intermediate; Markstein 1990); this test checks it
empirically against IEEE division on 10^8 operand pairs on the host (same IEEE-754 binary64 FMA semantics as v_fma_f64), and the
GPU tests run both builds of the kernel (tabulated reciprocals / v_div sequence) against the oracle."""
import os
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_markstein_sequence_equals_ieee_division():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "fastdiv_check")
        subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(HERE, "native", "fastdiv_check.c"), "-o", exe, "-lm"])
        out = subprocess.check_output([exe, "100000000", "17"], timeout=300).decode()
    assert "mismatches=0 (sign-of-zero only: 0)" in out, out


@pytest.mark.gpu
def test_ieee_division_build_of_the_fill_kernel_matches_oracle(monkeypatch):
    """PORESEQ_EXACT_DIV=1 selects the v_div_* build (used when a divisor is not a sane number): same bits"""
    import copy
    import numpy as np
    import backends as B
    from poreseq_amd import synth
    from poreseq_amd.poreseqcpp import PSAlign
    from poreseq_amd.util import DEFAULT_PARAMS
    P = dict(DEFAULT_PARAMS, verbose=0)
    draft, events, truth = synth.make_region(700, 5, 95, B.oracle_swalign, P)
    want = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P).ScorePoints()
    monkeypatch.setenv("PORESEQ_EXACT_DIV", "1")
    got = B.make_pa(PSAlign, draft, copy.deepcopy(events), P).ScorePoints()
    assert np.array_equal(np.array([m.score for m in got]), np.array([m.score for m in want]))
    # a divisor outside the sane range switches the AlignData to this build by itself
    monkeypatch.delenv("PORESEQ_EXACT_DIV")
    odd = copy.deepcopy(events)
    odd[1].stdv[5] = 1e-120
    a = B.make_pa(PSAlign, draft, copy.deepcopy(odd), P).ScoreEvents()
    b = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(odd), P).ScoreEvents()
    assert a == b
