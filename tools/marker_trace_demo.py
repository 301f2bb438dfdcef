#!/usr/bin/env python
"""Three records of the headline configuration with ROCTx ranges on (AEROBULK_AMD_ROCTX=1), for
    rocprofv3 --marker-trace --kernel-trace --stats -d <dir> -o trace -- python3 tools/marker_trace_demo.py
Shows the library's ranges (ab_session_init_stats, ab_session_compute) next to its kernels in one trace."""
import os
import sys

os.environ["AEROBULK_AMD_ROCTX"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402

IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def main():
    ni, nj, nt = 4320, 3600, 3
    f = ab.synth_fields_device(ni, nj, with_rad=True)
    with ab.Session("coare3p6", ni, nj, nt, True) as s:
        s.init(*[f[k] for k in IN6], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        for jt in range(1, nt + 1):
            s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
    print("done")


if __name__ == "__main__":
    main()
