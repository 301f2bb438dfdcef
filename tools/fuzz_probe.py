#!/usr/bin/env python
"""The values of a fuzz case (tests/test_gpu_fuzz.py) that need the backward clause: HIP error against the oracle's response to
single-input moves of 1, 2, 3, 4 and 8 ulp (and to FMA contraction).  Run on the GPU box.

    python tools/fuzz_probe.py seed algo skin zt zu niter [seed algo skin zt zu niter ...]
    python tools/fuzz_probe.py --range first_seed last_seed      (all nine configurations; prints the values with err > 2 S)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aerobulk_amd as ab  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_gpu_fuzz import _fields  # noqa: E402

NFLAG = [0, 0]
WORST = [0.0, 0.0]     # largest err / S(max over inputs, 8 ulp), err / S(sum over inputs, 8 ulp)
THRESH = 2.0
IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT = (("QL", "ql"), ("QH", "qh"), ("Tau_x", "tau_x"), ("Tau_y", "tau_y"), ("Evap", "evap"), ("T_s", "t_s"))


def run_oracle(algo, skin, zt, zu, niter, nt, f, variant=None):
    s = po.OracleSession(algo, f["sst"].size, nt, skin, variant=variant)
    return [s.compute(jt, zt, zu, niter, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"] if skin else None,
                      rad_lw=f["rad_lw"] if skin else None) for jt in range(1, nt + 1)]


def move(x, k):
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return x


CONFIGS = [("coare3p6", 1, 2.0, 10.0, 5), ("coare3p6", 0, 10.0, 10.0, 8), ("coare3p0", 1, 3.5, 17.0, 4), ("ecmwf", 1, 2.0, 10.0, 6),
           ("ecmwf", 0, 2.0, 10.0, 5), ("ncar", 0, 2.0, 10.0, 5), ("andreas", 0, 8.0, 12.0, 7), ("coare3p6", 1, 18.0, 25.0, 5),
           ("ncar", 0, 30.0, 10.0, 6)]


def main():
    a = sys.argv[1:]
    global THRESH
    if a and a[0] == "--thresh":       # print the values with err > thresh * S(one input, 8 ulp) (default 2)
        THRESH = float(a[1])
        a = a[2:]
    if a and a[0] == "--range":        # every configuration of tests/test_gpu_fuzz.py for the seeds first .. last-1
        a = [str(x) for seed in range(int(a[1]), int(a[2])) for c in CONFIGS for x in (seed, *c)]
    for c in range(0, len(a), 6):
        seed, algo, skin, zt, zu, niter = int(a[c]), a[c + 1], a[c + 2] == "1", float(a[c + 3]), float(a[c + 4]), int(a[c + 5])
        n = 60000 + 13 * seed
        f = _fields(seed, n)
        if seed % 2:
            keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
            f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
        nt = 3 if skin else 1
        ref = run_oracle(algo, skin, zt, zu, niter, nt, f)
        if any(r["rc"] for r in ref):
            continue                   # a record aborts on tau > 10 N/m2 (both sides: tests/test_gpu_fuzz.py)
        with ab.Session(algo, f["sst"].size, 1, nt, skin) as s:
            got = [s.compute(jt, zt, zu, *[f[k] for k in IN8[:6]], Niter=niter, rad_sw=f["rad_sw"] if skin else None,
                             rad_lw=f["rad_lw"] if skin else None) for jt in range(1, nt + 1)]
        NFLAG[0] += 0
        for jt in range(nt):
            for kg, kr in OUT if skin else OUT[:5]:
                r, g = ref[jt][kr], got[jt][kg]
                err = np.abs(g - r)
                bad = np.nonzero(err > 1e-10 * np.maximum(np.abs(r), 1e-6 * np.abs(r).max()))[0]
                NFLAG[1] += r.size
                if not bad.size:
                    continue
                NFLAG[0] += bad.size
                sub = {k: np.ascontiguousarray(v[bad]) for k, v in f.items()}
                base = run_oracle(algo, skin, zt, zu, niter, nt, sub)[jt][kr]
                sfma = np.abs(run_oracle(algo, skin, zt, zu, niter, nt, sub, variant="fma")[jt][kr] - base)
                S, per8 = {}, {}
                for ulp in (1, 2, 3, 4, 8):
                    acc = sfma.copy()
                    for k in IN8[:8 if skin else 6]:
                        rk = np.zeros_like(acc)
                        for sg in (1, -1):
                            p = dict(sub, **{k: move(sub[k], sg * ulp)})
                            rk = np.maximum(rk, np.abs(run_oracle(algo, skin, zt, zu, niter, nt, p)[jt][kr] - base))
                        acc = np.maximum(acc, rk)
                        per8[k] = np.maximum(per8.get(k, 0.0), rk)      # response to moves of <= 8 ulp of input k alone
                    S[ulp] = acc
                cum = {u: np.maximum.reduce([S[v] for v in S if v <= u]) for u in S}
                ssum = np.sum([per8[k] for k in per8], axis=0)           # all inputs moving at once (first order)
                for i, b in enumerate(bad):
                    rat = {u: err[b] / max(cum[u][i], 1e-300) for u in cum}
                    WORST[0] = max(WORST[0], rat[8])
                    WORST[1] = max(WORST[1], err[b] / max(ssum[i], 1e-300))
                    if rat[1] > THRESH:
                        print(f"seed {seed} {algo} skin={int(skin)} zt={zt} jt={jt + 1} {kr}[{b}]: ref {r[b]:.6e} err {err[b]:.3e} "
                              f"err/S(moves <= 1,2,3,4,8 ulp) = " + " ".join(f"{rat[u]:.2f}" for u in (1, 2, 3, 4, 8))
                              + f" err/sum over inputs (8 ulp) = {err[b] / max(ssum[i], 1e-300):.2f}"
                              + " | response per input (8 ulp)/err: " + " ".join(f"{k}={per8[k][i] / err[b]:.2f}" for k in per8)
                              + f" fma={sfma[i] / err[b]:.2f}"
                              + f" | sst {f['sst'][b]:.3f} t {f['t_zt'][b]:.3f} q {f['hum_zt'][b]:.5f} wind {np.hypot(f['u_zu'][b], f['v_zu'][b]):.3f}"
                              + (f" rsw {f['rad_sw'][b]:.1f}" if skin else ""), flush=True)


if __name__ == "__main__":
    main()
    print(f"values compared {NFLAG[1]}, beyond the forward bar {NFLAG[0]}; largest err / response to moves of <= 8 ulp: "
          f"one input {WORST[0]:.2f}, all inputs {WORST[1]:.2f}")
