"""TURB_NEUTRAL_10M (mod_blk_neutral_10m.f90:33): golden data tests/golden/neutral10.npz from the unmodified
reference (tools/gen_neutral10_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

G = dict(np.load(os.path.join(GOLDEN, "neutral10.npz")))
CASES = [("coare3p0", 5), ("coare3p6", 5), ("coare3p6", 2), ("ecmwf", 8), ("ncar", 5)]
OUT = ("CdN10", "ChN10", "CeN10", "z0")
FDRV = os.path.join(ROOT, "aerobulk_amd", "fortran", "neutral10_driver.x")


@pytest.mark.parametrize("algo,niter", CASES)
def test_oracle_neutral10_matches_reference(oracle, algo, niter):
    o = oracle.oracle_neutral10(algo, niter, G["U_N10"])
    for k in OUT:
        np.testing.assert_allclose(o[k], G[f"{algo}_n{niter}_{k}"], rtol=1e-15, atol=0, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("algo,niter", CASES)
def test_hip_neutral10_matches_reference(oracle, algo, niter):
    import torch
    import aerobulk_amd as ab
    host = ab.turb_neutral_10m(algo, G["U_N10"], nb_iter=niter)
    dev = ab.turb_neutral_10m(algo, torch.from_numpy(G["U_N10"]).cuda(), nb_iter=niter)
    for k in OUT:
        np.testing.assert_allclose(host[k], G[f"{algo}_n{niter}_{k}"], rtol=1e-12, atol=0, err_msg=k)
        np.testing.assert_array_equal(dev[k].cpu().numpy(), host[k])
    if os.path.exists(FDRV):
        f = oracle.run_neutral10_driver(FDRV, algo, niter, G["U_N10"])
        for k in OUT:
            np.testing.assert_array_equal(f[k], host[k], err_msg=k + " [fortran]")


@pytest.mark.gpu
def test_neutral10_andreas_is_refused_like_the_reference():
    import aerobulk_amd as ab
    with pytest.raises(ab.AerobulkError):
        ab.turb_neutral_10m("andreas", G["U_N10"])
