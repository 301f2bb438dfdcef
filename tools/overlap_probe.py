#!/usr/bin/env python
"""What NOT ending the launch would buy on a 4320 x 450 slab (profiles/r5_notes.md §8, first lead): consecutive records of TWO independent sessions on two
streams, so that the workgroups of record t + 1 take each CU the moment record t's persistent workgroup leaves it.  Only callers whose consecutive records do
not depend on one another (ensembles, several regions) could use it — a time loop with the warm layer's state cannot; informational, never `value`.
    python tools/overlap_probe.py      (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def main():
    dev = torch.device("cuda", 0)
    with ab.Session("coare3p6", 4320, 3600, 1, False) as s:      # clock ramp
        f = ab.synth_fields_device(4320, 3600)
        for _ in range(60):
            s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, check=False)
        torch.cuda.synchronize()
    for nj in (450, 900, 3600):
        f = ab.synth_fields_device(4320, 3600, 0, nj)
        n = 4320 * nj
        outs = [{k: torch.empty(n, dtype=torch.float64, device=dev) for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")} for _ in range(2)]
        st = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        with ab.Session("coare3p6", 4320, nj, 1, True) as sa, ab.Session("coare3p6", 4320, nj, 1, True) as sb:
            ss = (sa, sb)

            def run(k, which):
                kw = dict(Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"], out=outs[which], check=False)
                with torch.cuda.stream(st[which]):
                    ss[which].compute(1, 2.0, 10.0, *[f[k_] for k_ in IN6], **kw)

            def timed(two, reps=80):
                best = 1e9
                for _ in range(4):
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    st[0].wait_event(e0); st[1].wait_event(e0)
                    for i in range(reps):
                        run(i, (i & 1) if two else 0)
                    torch.cuda.current_stream().wait_stream(st[0]); torch.cuda.current_stream().wait_stream(st[1])
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / reps)
                return best
            for i in range(10):
                run(i, i & 1)
            one, two = timed(False), timed(True)
            same = all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0])
        print(f"{nj:5d} rows: one stream {one:.4f} ms per record, two streams (records of two sessions alternating) {two:.4f} ms per record ({one / two:.3f} x); same bits: {same}")


if __name__ == "__main__":
    main()
