#!/usr/bin/env python
"""Are the outputs of two builds of the library bit-identical?  Each library (build/var/libab_<tag>.so; `cur` = the in-tree one) runs the
same configurations on the same synthetic fields in its own process; the SHA-1 of every output field is compared.

    python tools/lib_identical.py [--grid 1440x1080] [--configs coare3p6:1:5,ecmwf:1:5] [--precision f64] cur newtag
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
ni, nj = (int(x) for x in sys.argv[2].split("x"))
prec = sys.argv[4]
f = ab.synth_fields_device(ni, nj, precision=prec)
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
out = {}
for cfg in sys.argv[3].split(","):
    algo, skin, niter = cfg.split(":")
    skin, niter = skin == "1", int(niter)
    with ab.Session(algo, ni, nj, 3, skin, precision=prec) as s:
        d = {}
        for jt in (1, 2, 3):                  # three records: the warm layer's state carries over
            o = s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=niter, rad_sw=f["rad_sw"] if skin else None,
                          rad_lw=f["rad_lw"] if skin else None, check=False)
            for k, v in o.items():
                d[f"{k}@{jt}"] = hashlib.sha1(v.cpu().numpy().tobytes()).hexdigest()
        out[cfg] = d
print("RESULT " + json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs=2)
    ap.add_argument("--grid", default="1440x1080")
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--configs", default="coare3p6:1:5,coare3p0:1:5,coare3p6:1:8")
    a = ap.parse_args()
    res = []
    for t in a.tags:
        env = dict(os.environ)
        if t != "cur":
            env["AEROBULK_AMD_LIB"] = os.path.join(ROOT, "build", "var", f"libab_{t}.so")
        else:
            env.pop("AEROBULK_AMD_LIB", None)
        o = subprocess.run([sys.executable, "-c", CHILD, ROOT, a.grid, a.configs, a.precision], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            raise SystemExit(f"{t}: {o.stdout[-1500:]}{o.stderr[-3000:]}")
        res.append(json.loads(line[0][7:]))
    ok = True
    for cfg in res[0]:
        diff = [k for k in res[0][cfg] if res[0][cfg][k] != res[1][cfg][k]]
        print(f"{cfg}: {'identical' if not diff else 'DIFFERENT in ' + ', '.join(diff)} ({len(res[0][cfg])} fields x records)")
        ok &= not diff
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
