#!/usr/bin/env python
"""Kernel time of the headline configuration (COARE3p6 + skin, nb_iter 5) on row slabs of the 4320-wide benchmark grid — what one rank of
N owns — for the CU-wide persistent kernel and the 256-thread block kernel (AEROBULK_AMD_CU_KERNEL=1 / 0), same box, interleaved passes,
each in a fresh process.  Two clocks per slab:
   single : HIP events around ONE launch (ab_session_last_kernel_ms), median of 15 — includes what a lone dispatch costs
   piped  : 40 launches enqueued back to back between two events, divided by 40 — what a time loop pays per record
Rate per cell relative to the full grid is the figure the 8-GPU scaling estimate needs.

    python tools/slab_rates.py [--passes 3] [--rows 225,450,900,1800,3600] [tags ...]      (GPU box; tags: build/var/libab_<tag>.so, `cur`;
    tag@ENV=VALUE runs that library with the environment variable set, e.g. cur@AEROBULK_AMD_TAIL_X=2)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import torch
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
rows = [int(x) for x in sys.argv[2].split(",")]
algo, niter = sys.argv[3], int(sys.argv[4])
out = {}
with ab.Session("coare3p6", 4320, 3600, 1, False) as s:      # clock ramp
    f = ab.synth_fields_device(4320, 3600)
    for _ in range(60):
        s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
    s.last_kernel_ms()
for nj in rows:
    f = ab.synth_fields_device(4320, 3600, 0, nj)
    with ab.Session(algo, 4320, nj, 1, True) as s:
        kw = dict(Niter=niter, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], check=False)
        ms = []
        for _ in range(15):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], **kw)
            ms.append(s.last_kernel_ms())
        ms.sort()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(40):
                s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        out[str(nj)] = [ms[len(ms) // 2], best]
print("RESULT " + json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="*", default=["cur"])
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--rows", default="28,113,225,450,900,1800,3600")
    ap.add_argument("--algo", default="coare3p6")
    ap.add_argument("--niter", type=int, default=5)
    ap.add_argument("--kernels", default="1,0")
    a = ap.parse_args()
    variants = [(t, m) for t in a.tags for m in a.kernels.split(",")]
    res = {v: [] for v in variants}
    for p in range(a.passes):
        for v in (variants if p % 2 == 0 else variants[::-1]):
            t, mode = v
            e = dict(os.environ, AEROBULK_AMD_CU_KERNEL=mode)
            t, *envs = t.split("@")                          # tag@ENV=VALUE@...: the same library under other environment variables
            for kv in envs:
                e[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            if t != "cur":
                e["AEROBULK_AMD_LIB"] = os.path.join(ROOT, "build", "var", f"libab_{t}.so")
            else:
                e.pop("AEROBULK_AMD_LIB", None)
            pr = subprocess.run([sys.executable, "-c", CHILD, ROOT, a.rows, a.algo, str(a.niter)], env=e, capture_output=True, text=True)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(v, "FAILED", pr.stdout[-500:], pr.stderr[-1500:])
                continue
            res[v].append(json.loads(line[-1][7:]))
    med = lambda xs: sorted(xs)[len(xs) // 2]
    rows = a.rows.split(",")
    for t, mode in variants:
        r = res[(t, mode)]
        if not r:
            continue
        name = f"{t} / " + ("CU kernel (forced)" if mode == "1" else "block kernel")
        full = med([x[rows[-1]][1] for x in r]) / (4320 * int(rows[-1]))
        print(f"--- {name}: {a.algo} + skin, nb_iter {a.niter}")
        print(f"{'rows':>6s} {'cells':>10s} {'single [ms]':>12s} {'piped [ms]':>11s} {'Mcell/s':>9s} {'rate vs ' + rows[-1] + ' rows':>18s}")
        for nj in rows:
            s1, pp = med([x[nj][0] for x in r]), med([x[nj][1] for x in r])
            n = 4320 * int(nj)
            print(f"{nj:>6s} {n:10d} {s1:12.4f} {pp:11.4f} {n / pp / 1e3:9.0f} {full * n / pp:18.3f}")
    print("passes:", json.dumps({f"{t}:{m}": v for (t, m), v in res.items()}))


if __name__ == "__main__":
    main()
