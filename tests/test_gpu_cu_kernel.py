"""GPU: flux_kernel_cu — one workgroup of sixteen waves per CU, four teams with barriers of their own, every LDS table once per CU
(ab_kernels.hip; serves fp64 COARE 3.0 / 3.6 with the skin schemes on grids that fill the chip).  It evaluates the same
polynomials as the 256-thread flux_kernel (which reads the long tables through L1), so:
  * forced on (AEROBULK_AMD_CU_KERNEL=1) it reproduces the reference's golden vectors like any other kernel;
  * forced on and forced off (=0) give the same BITS, on ragged grids, over three records with the warm-layer state carried,
    for fp64 and fp32 arrays (AB_F32_STORAGE), with regrouping on and off — which is also what keeps a j-block computed alone
    (small grid: 256-thread kernel) bit-identical to the same rows of a full-grid launch (CU kernel)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
out = {}
for algo, ni, nj, prec, regroup, niter in (("coare3p6", 1013, 311, "f64", 1, 5), ("coare3p0", 640, 97, "f64", 1, 4), ("coare3p6", 777, 130, "f32_storage", 1, 5),
                                           ("coare3p6", 512, 64, "f64", 0, 8), ("coare3p6", 5, 1, "f64", 1, 5),
                                           # the persistent loop's steady state: 2.2 M cells = four or more tiles per team (phase 4 -> phase 1 with no
                                           # barrier, the look-ahead of the tile queue, the counters re-armed across the launches of one session)
                                           ("coare3p6", 2161, 1019, "f64", 1, 2)):
    f = ab.synth_fields_device(ni, nj, precision="f64" if prec == "f64" else "f32")
    with ab.Session(algo, ni, nj, 3, True, precision=prec) as s:
        s.set_regroup(bool(regroup))
        for jt in (1, 2, 3):
            o = s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=niter, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
            for k, v in o.items():
                out[f"{algo}/{ni}x{nj}/{prec}/{regroup}/{jt}/{k}"] = hashlib.sha1(v.cpu().numpy().tobytes()).hexdigest()
        st = s.wl_state()
        out[f"{algo}/{ni}x{nj}/{prec}/{regroup}/wl"] = hashlib.sha1(np.concatenate([st[k] for k in sorted(st)]).tobytes()).hexdigest()
print("RESULT " + json.dumps(out))
'''


def _run(mode):
    e = dict(os.environ, AEROBULK_AMD_CU_KERNEL=mode)
    pr = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
    return json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


def test_cu_kernel_gives_the_bits_of_the_block_kernel():
    on, off = _run("1"), _run("0")
    assert on.keys() == off.keys() and len(on) == 6 * (3 * 6 + 1)
    diff = [k for k in on if on[k] != off[k]]
    assert not diff, diff[:10]


def test_cu_kernel_reproduces_the_golden_vectors():
    """tests/test_gpu_golden.py (the reference's own vectors) and test_gpu_parity.py with the CU-wide kernel forced on for the 2 048-cell sweeps."""
    e = dict(os.environ, AEROBULK_AMD_CU_KERNEL="1")
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_golden.py"), os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                         "-m", "gpu", "-q", "-x", "-k", "coare", "-p", "no:cacheprovider"], env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-3000:] + pr.stderr[-2000:]
    assert " passed" in pr.stdout
