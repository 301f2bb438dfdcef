#!/usr/bin/env python
"""Kernel time of the helper kernels (ab_phymbl, include/aerobulk_amd.h) on device-resident arrays of the benchmark grid's size: run under
rocprofv3 --kernel-trace --stats, the per-kernel averages against the algorithmic bytes (8 B per operand and cell) give their HBM rate.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_phymbl -o ph -- python3 tools/phymbl_bench.py        (GPU box)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
import aerobulk_amd as ab  # noqa: E402
import phymbl_cases as pc  # noqa: E402

N = 4320 * 3600
NAMES = ("virt_temp", "q_sat", "theta_from_z", "rho_air", "one_on_l", "ri_bulk", "bf_qlat", "uqt_qns", "psi_m_coare", "psi_h_ecmwf", "fg_us", "e_sat", "visc_air")


def main():
    cols = pc.make_columns(n=4096)
    dev = {k: torch.tensor(cols[i], device="cuda").repeat(N // 4096 + 1)[:N].contiguous() for i, k in enumerate(pc.COLUMNS)}
    for name in NAMES:
        fn, par0, flag, ins, oi = pc.CALLS[name]
        args = [None if c is None else dev[c] for c in ins]
        for _ in range(6):
            ab.phymbl(fn, args, par0, flag, pc.N_OUT.get(fn, 1), par1=pc.PAR1.get(fn, 0.))
        torch.cuda.synchronize()
        nin, nout = sum(a is not None for a in args), pc.N_OUT.get(fn, 1)
        print(f"{name}: fn {fn}, {nin} in, {nout} out, {8 * (nin + nout)} B per cell, {8 * (nin + nout) * N / 1e6:.0f} MB per launch")


if __name__ == "__main__":
    main()
