#!/usr/bin/env python
"""Which cells differ between flux_kernel_cu (forced) and the 256-thread flux_kernel on one grid: indices, tiles, positions in the tile.
    python tools/cu_mismatch_probe.py [algo ni nj niter]      (GPU box)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
algo, ni, nj, niter = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
f = ab.synth_fields_device(ni, nj)
with ab.Session(algo, ni, nj, 1, True) as s:
    outs = []
    for rep in range(3):
        o = s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=niter, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        outs.append(np.stack([o[k].cpu().numpy() for k in sorted(o)]))
np.save(sys.argv[6], np.stack(outs))
"""
algo, ni, nj, niter = (sys.argv[1:5] + ["coare3p0", "1000", "565", "1"][len(sys.argv) - 1:])[:4] if len(sys.argv) > 1 else ("coare3p0", "1000", "565", "1")
res = {}
with tempfile.TemporaryDirectory() as d:
    for mode in ("1", "0"):
        out = os.path.join(d, f"o{mode}.npy")
        e = dict(os.environ, AEROBULK_AMD_CU_KERNEL=mode)
        pr = subprocess.run([sys.executable, "-c", CHILD, ROOT, algo, ni, nj, niter, out], env=e, capture_output=True, text=True)
        if pr.returncode:
            raise SystemExit(pr.stdout[-800:] + pr.stderr[-2500:])
        res[mode] = np.load(out)
a, b = res["1"], res["0"]
n = a.shape[2]
print(f"{algo} {ni}x{nj} niter {niter}: n = {n}; block kernel reps identical: {all(np.array_equal(b[0], b[i]) for i in range(3))}")
tail = min(n, 1024 * 256)
nfull = (n - tail) // 512
for rep in range(3):
    bad = np.unique(np.nonzero(a[rep] != b[0])[1])
    print(f"  rep {rep}: {bad.size} cells differ")
    if bad.size:
        t = np.where(bad < nfull * 512, bad // 512, nfull + (bad - nfull * 512) // 256)
        pos = np.where(bad < nfull * 512, bad % 512, (bad - nfull * 512) % 256)
        tiles, cnt = np.unique(t, return_counts=True)
        print(f"    nfull {nfull}, tiles with differences: {tiles.size}: first {tiles[:12].tolist()} counts {cnt[:12].tolist()}")
        print(f"    positions in tile (first 24): {pos[:24].tolist()}")
        k = bad[0]
        print(f"    first bad cell {k}: cu {a[rep][:, k].tolist()} block {b[0][:, k].tolist()}")
