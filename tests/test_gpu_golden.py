"""GPU: HIP kernels through the C ABI vs the golden vectors produced by the UNMODIFIED reference
(tests/golden, tools/gen_golden.py): all 5 algorithms x {no skin, skin} x nb_iter {1,5,8} x zt {2,10}
x humidity {sh,rh,dp}, and 3-record warm-layer carry-over."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_hot_parity, load_golden_case, load_manifest, sensitivity

pytestmark = pytest.mark.gpu
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
CAP = {"ql": "QL", "qh": "QH", "tau_x": "Tau_x", "tau_y": "Tau_y", "evap": "Evap", "t_s": "T_s"}


@pytest.mark.parametrize("case", load_manifest(), ids=lambda c: c["name"])
def test_hip_matches_reference_golden(case, oracle):
    import aerobulk_amd as ab
    inp, recs, keys = load_golden_case(case)
    n = inp["sst"].size
    sens = sensitivity(oracle, case["algo"], case["skin"], case["zt"], case["zu"], case["niter"], inp, nt=case["nt"],
                       hum_type=case["hum_type"])
    with ab.Session(case["algo"], n, 1, case["nt"], case["skin"]) as s:
        rep = s.init(*[inp[k] for k in IN6], rad_sw=inp["rad_lw"] if case["skin"] else None,
                     rad_lw=inp["rad_lw"] if case["skin"] else None)
        assert rep["hum_type"] == case["hum_type"]  # type_of_humidity detection on the GPU
        for jt, ref in enumerate(recs, 1):
            got = s.compute(jt, case["zt"], case["zu"], *[inp[k] for k in IN6], Niter=case["niter"],
                            rad_sw=inp["rad_sw"] if case["skin"] else None, rad_lw=inp["rad_lw"] if case["skin"] else None)
            got = {k: got[CAP[k]] for k in keys}
            assert_hot_parity(got, ref, keys, sens=sens, jt=jt, label=f"{case['name']} jt={jt}")


def test_hip_reproduces_17_digit_pins():
    import aerobulk_amd as ab
    p = json.load(open(os.path.join(GOLDEN, "pins_2cell.json")))
    f = {k: np.array(v, dtype=np.float64) for k, v in p["inputs"].items()}
    for name, outs in p["outputs"].items():
        algo, sk = name.rsplit("_", 1)
        skin = sk == "skin"
        with ab.Session(algo, 2, 1, 1, skin) as s:
            o = s.compute(1, p["zt"], p["zu"], *[f[k] for k in IN6], Niter=p["niter"],
                          rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
        for k, hexes in outs.items():
            ref = np.array([float.fromhex(h) for h in hexes])
            np.testing.assert_allclose(o[CAP[k]], ref, rtol=1e-11, atol=0, err_msg=f"{name} {k}")
