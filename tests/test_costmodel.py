"""tools/costmodel.cpp (the product's ab_physics.hpp compiled for the host with an instrumented scalar: where the VALU issue slots go,
DESIGN.md §3.1) still builds against the current headers and reproduces the oracle's fluxes on its sample — it is the development
aid the kernel work is steered with, and it breaks silently when a template signature in the headers changes."""
import os
import re
import subprocess

from conftest import ROOT


def test_cost_model_builds_runs_and_matches_the_oracle(tmp_path, oracle):
    exe = str(tmp_path / "costmodel")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-o", exe, os.path.join(ROOT, "tools", "costmodel.cpp")])
    out = subprocess.check_output([exe], text=True)
    m = re.search(r"coare3p6 skin=1 nb_iter=5, (\d+) cells: (\d+) slots per cell \(sum QL ([-+0-9.e]+)\)", out)
    assert m, out[:300]
    ncell, slots, sum_ql = int(m.group(1)), int(m.group(2)), float(m.group(3))
    assert 3000 < slots < 6500
    # the same cells through the oracle (the model samples the 4320 x 3600 benchmark fields at i = 6, 12, ..., j = 10, 20, ...):
    # the model executes the real arithmetic, so the checksum must agree
    import numpy as np
    assert ncell == 720 * 360
    IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
    rows = [oracle.synth_fields(4320, 3600, 10 * j - 1, 1) for j in range(1, 361)]
    f = {k: np.concatenate([r[k][5::6] for r in rows]) for k in rows[0]}
    o = oracle.OracleSession("coare3p6", ncell, 1, True)
    ref = o.compute(1, 2.0, 10.0, 5, *[f[k] for k in IN6], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
    assert abs(ref["ql"].sum() - sum_ql) <= 1e-9 * abs(sum_ql)
    for region in ("cool_skin", "psi_coare", "wl_coare", "q_sat"):
        assert region in out
