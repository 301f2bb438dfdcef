#!/usr/bin/env python
"""VERDICT r5 item 2, lever (a): "two kernels, one record" — the persistent flux_kernel_cu on the first rows of a 4320 x 450 slab and, on a second
stream, 256-thread flux_kernel blocks on the rest, which the dispatcher places on a CU the moment its persistent workgroup has left.  Measured
WITHOUT touching the library: two sessions over the two row ranges of the same resident fields (the library picks flux_kernel_cu from ~1.6 M
cells on and flux_kernel below), one torch stream each; every record waits for BOTH kernels of the record before it (no overlap of consecutive
records — that would be another measurement).  Against ONE session over the 450 rows.      python tools/two_kernel_probe.py      (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
NI, NJ = 4320, 450


def main():
    dev = torch.device("cuda", 0)
    with ab.Session("coare3p6", 4320, 3600, 1, False) as s:      # clock ramp
        f = ab.synth_fields_device(4320, 3600)
        for _ in range(60):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
        torch.cuda.synchronize()
    f = ab.synth_fields_device(NI, 3600, 0, NJ)
    out = {k: torch.empty(NI * NJ, dtype=torch.float64, device=dev) for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")}
    res = {}

    def timed(fn, reps=60):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best

    with ab.Session("coare3p6", NI, NJ, 1, True) as s1:
        kw = dict(Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], out=out, check=False)
        one = lambda: s1.compute(1, 2.0, 10.0, *[f[k] for k in IN6], **kw)
        for _ in range(10):
            one()
        res["one kernel, 450 rows"] = timed(one)
        ref = {k: v.clone() for k, v in out.items()}
    st2 = torch.cuda.Stream(device=dev)
    for rows_cu in (430, 410, 390, 382):
        n1 = NI * rows_cu
        a = {k: v[:n1] for k, v in f.items()}
        b = {k: v[n1:] for k, v in f.items()}
        oa = {k: v[:n1] for k, v in out.items()}
        ob = {k: v[n1:] for k, v in out.items()}
        for v in out.values():
            v.zero_()
        with ab.Session("coare3p6", NI, rows_cu, 1, True) as sa, ab.Session("coare3p6", NI, NJ - rows_cu, 1, True) as sb:
            def two():
                cur = torch.cuda.current_stream()
                st2.wait_stream(cur)                     # the record before is complete (both kernels) before either kernel of this one starts
                sa.compute(1, 2.0, 10.0, *[a[k] for k in IN6], Niter=5, rad_sw=a["rad_sw"], rad_lw=a["rad_lw"], out=oa, check=False)
                with torch.cuda.stream(st2):
                    sb.compute(1, 2.0, 10.0, *[b[k] for k in IN6], Niter=5, rad_sw=b["rad_sw"], rad_lw=b["rad_lw"], out=ob, check=False)
                cur.wait_stream(st2)
            for _ in range(10):
                two()
            t = timed(two)
            torch.cuda.synchronize()
            same = all(torch.equal(out[k], ref[k]) for k in out)
        res[f"two kernels: flux_kernel_cu on {rows_cu} rows + flux_kernel on {NJ - rows_cu} rows, second stream"] = t
        res[f"  bit-identical to the one-kernel record ({rows_cu})"] = same
    for k, v in res.items():
        print(f"{k:95s} {v:.4f} ms" if isinstance(v, float) else f"{k:95s} {v}")
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
