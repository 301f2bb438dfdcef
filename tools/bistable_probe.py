#!/usr/bin/env python
"""The values of ONE fuzz case (tests/test_gpu_fuzz.py) beyond the forward bar, with every digit: cell, record, field, oracle value, HIP
value — for the in-tree library and for any build/var/libab_<tag>.so given (each in its own process).  Run on the GPU box; the cells
found go to tools/gen_bistable.py (build container), which asks the reference's own builds.

    python tools/bistable_probe.py seed algo skin zt zu niter [tag ...]  > gpurun_out/bistable_<seed>.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import aerobulk_amd as ab
from oracle import pyoracle as po
from test_gpu_fuzz import _fields
seed, algo, skin, zt, zu, niter = int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1", float(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT = (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s"))
f = _fields(seed, 60000 + 13 * seed)
if seed % 2:
    keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
    f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
nt = 3 if skin else 1
rad = dict(rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
o = po.OracleSession(algo, f["sst"].size, nt, skin)
out = []
with ab.Session(algo, f["sst"].size, 1, nt, skin) as s:
    for jt in range(1, nt + 1):
        ref = o.compute(jt, zt, zu, niter, *[f[k] for k in IN8[:6]], **rad)
        got = s.compute(jt, zt, zu, *[f[k] for k in IN8[:6]], Niter=niter, **rad)
        for kg, kr in (OUT if skin else OUT[:5]):
            r, g = ref[kr], got[kg]
            bad = np.nonzero(np.abs(g - r) > 1e-10 * np.maximum(np.abs(r), 1e-6 * np.abs(r).max()))[0]
            for b in bad:
                out.append(dict(jt=jt, field=kr, cell=int(b), ref=float(r[b]).hex(), got=float(g[b]).hex(), ref_f=float(r[b]), got_f=float(g[b]),
                                rel=float(abs(g[b] - r[b]) / abs(r[b])), inputs={k: float(f[k][b]) for k in IN8}))
print(json.dumps(out))
"""


def main():
    a = sys.argv[1:]
    case, tags = a[:6], a[6:]
    res = {}
    for tag in ["cur"] + tags:
        env = dict(os.environ)
        if tag != "cur":
            env["AEROBULK_AMD_LIB"] = os.path.join(ROOT, "build", "var", f"libab_{tag}.so")
        p = subprocess.run([sys.executable, "-c", CHILD, ROOT, *case], env=env, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr[-2000:])
            raise SystemExit(p.returncode)
        res[tag] = json.loads(p.stdout.strip().split("\n")[-1])
    print(json.dumps(dict(case=case, values=res), indent=1))


if __name__ == "__main__":
    main()
