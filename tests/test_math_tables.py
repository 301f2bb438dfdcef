"""The LDS tables of the fp64 log / exp (aerobulk_amd/csrc/ab_fastmath.hpp) are what tools/gen_logtab.py / gen_exptab.py define:
invc[k] = double(64/k), logc[k] = -log(invc[k]) of that rounded value (k = 45..91; k = 64 holds exactly (1, 0)), T[j] = 2^(j/32)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

mp = pytest.importorskip("mpmath")
HDR = open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_fastmath.hpp")).read()


def _table(name):
    m = re.search(rf"AB_TAB double {name}\[[^\]]*\] = \{{(.*?)\}};", HDR, re.S)
    assert m, name
    return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])


def test_log_table_is_exactly_what_the_identity_needs():
    mp.mp.dps = 50
    t = _table("kLogTab").reshape(-1, 2)
    assert t.shape == (47, 2)
    for k, (invc, logc) in zip(range(45, 92), t):
        assert invc == float(mp.mpf(64) / k)
        assert logc == float(-mp.log(mp.mpf(invc)))           # of the ROUNDED reciprocal: log m = log(m invc) - log(invc) exactly
    assert tuple(t[64 - 45]) == (1.0, 0.0)


def test_exp_table_and_reduction_constants():
    mp.mp.dps = 50
    t = _table("kExpTab")
    assert t.size == 32
    assert all(t[j] == float(mp.mpf(2) ** (mp.mpf(j) / 32)) for j in range(32))
    # the 30-bit heads of ln2/32 and log10(2)/32 times any |k| < 2^22 are exact in double; head + tail reproduce the constant
    for head, tail, val in ((0.021660849393811077, -1.312785960212839e-12, mp.log(2) / 32),
                            (0.009407187360920943, 3.5784690306318245e-12, mp.log10(2) / 32)):
        assert f"{head!r}" in HDR and f"{tail!r}" in HDR
        m, _ = np.frexp(head)
        assert (m * 2.0 ** 30) == int(m * 2.0 ** 30)
        assert abs(mp.mpf(head) + mp.mpf(tail) - val) < mp.mpf(10) ** -28
