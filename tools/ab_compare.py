#!/usr/bin/env python
"""Same-box A/B of kernel variants: every library given (build/var/libab_<tag>.so, tools/build_variant.sh; `cur` = the in-tree
library) times the same configurations in interleaved passes, each pass in a fresh process (AEROBULK_AMD_LIB selects the library).
Box-to-box variation of one binary is about +-3 %, same-box repeatability about +-0.5 %: variants are only comparable this way.

    python tools/ab_compare.py [--passes 3] [--grid 4320x3600] [--configs coare3p6:1:5,coare3p6:0:8,ecmwf:1:5] tag1 tag2 ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
ni, nj = (int(x) for x in sys.argv[2].split("x"))
cfgs = [c.split(":") for c in sys.argv[3].split(",")]
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
prec = sys.argv[4]
f = ab.synth_fields_device(ni, nj, precision=prec)
f64 = f if prec == "f64" else ab.synth_fields_device(ni, nj)
with ab.Session("coare3p6", ni, nj, 1, False) as s:      # clock ramp
    for _ in range(80):
        s.compute(1, 2.0, 10.0, *[f64[k] for k in IN6], Niter=5, check=False)
    s.last_kernel_ms()
out = {}
for algo, skin, niter in cfgs:
    skin, niter = skin == "1", int(niter)
    with ab.Session(algo, ni, nj, 1, skin, precision=prec) as s:
        ms = []
        for _ in range(12):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=niter, rad_sw=f["rad_sw"] if skin else None,
                      rad_lw=f["rad_lw"] if skin else None, check=False)
            ms.append(s.last_kernel_ms())
        ms.sort()
        out[f"{algo}:{int(skin)}:{niter}"] = ms[len(ms) // 2]
print("RESULT " + json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="+")
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--grid", default="4320x3600")
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--configs", default="coare3p6:1:5,coare3p6:0:8,ecmwf:1:5,coare3p0:1:5,andreas:0:5,ncar:0:5")
    a = ap.parse_args()
    res = {t: {} for t in a.tags}
    for p in range(a.passes):
        for t in (a.tags if p % 2 == 0 else a.tags[::-1]):     # A B | B A | ...: whatever drifts with time (clocks, temperature) hits both
            env = dict(os.environ)
            if t != "cur":
                env["AEROBULK_AMD_LIB"] = os.path.join(ROOT, "build", "var", f"libab_{t}.so")
            else:
                env.pop("AEROBULK_AMD_LIB", None)
            o = subprocess.run([sys.executable, "-c", CHILD, ROOT, a.grid, a.configs, a.precision], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(t, "FAILED", o.stderr[-500:])
                continue
            for k, v in json.loads(line[0][7:]).items():
                res[t].setdefault(k, []).append(v)
    cfgs = a.configs.split(",")
    print(f"{'config':18s} " + " ".join(f"{t:>22s}" for t in a.tags))
    base = a.tags[0]
    for c in cfgs:
        row = []
        for t in a.tags:
            v = sorted(res[t].get(c, [float('nan')]))
            med = sum(v) / len(v) if len(v) % 2 == 0 else v[len(v) // 2]      # even number of passes: the mean (order-balanced)
            b = sorted(res[base].get(c, [float('nan')]))
            bm = sum(b) / len(b) if len(b) % 2 == 0 else b[len(b) // 2]
            row.append(f"{med:8.3f} ms ({100 * (med / bm - 1):+5.1f} %)")
        print(f"{c:18s} " + " ".join(f"{r:>22s}" for r in row))
    print("passes:", json.dumps(res))


if __name__ == "__main__":
    main()
