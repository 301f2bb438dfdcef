"""CPU: the fp64 elementary-function algorithms of aerobulk_amd/csrc/ab_fastmath.hpp compiled for the host
(hardware seeds emulated at float accuracy) and measured against 80-bit libm: max error in ulp."""
import os
import subprocess

from conftest import ROOT

LIMITS = {"log_wide": 2.5, "log_near1": 2.5, "log10": 4.0, "exp": 1.5, "exp_small": 1.5, "exp10": 2.0, "exp10_small": 2.0,
          "atan": 2.5, "atan_wide": 2.5, "sqrt": 0.51, "rcp": 0.51, "div": 0.51, "cbrt": 8.0, "rcbrt": 4.0, "rqrt": 4.0, "rqrt_used": 1.5}


def test_fastmath_algorithms_on_host(tmp_path):
    exe = str(tmp_path / "fm_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "fastmath_host.cpp")])
    out = subprocess.check_output([exe], text=True)
    print(out)
    seen = {}
    for line in out.splitlines():
        p = line.split()
        if p[0] in LIMITS:
            seen[p[0]] = float(p[1])
    assert set(seen) == set(LIMITS)
    for k, v in seen.items():
        assert v <= LIMITS[k], (k, v)
    assert "exp(-1e4)=0 exp(1e4)=inf sqrt(0)=0 cbrt(0)=0" in out
