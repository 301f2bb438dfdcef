#!/usr/bin/env python
"""Kernel timing of the sea-ice algorithms on device-resident synthetic polar fields (4320x3600 cells, fp64, nb_iter=5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aerobulk_amd as ab  # noqa: E402


def main():
    n = 4320 * 3600
    k = torch.arange(n, device="cuda", dtype=torch.float64)
    r = lambda a, c: torch.frac(k * a + c)
    Ts = 233.15 + 40. * r(0.6180339887498949, 0.)
    tht = Ts - 6. + 16. * r(0.7548776662466927, 0.1)
    qs = 3.8e-3 * torch.exp(0.09 * (Ts - 273.15))
    q = (0.5 + 0.6 * r(0.3247179572447460, 0.3)) * 3.8e-3 * torch.exp(0.09 * (tht - 273.15))
    W = 0.05 + 24. * r(0.8191725133961645, 0.4) ** 2
    fri = torch.clamp(-0.05 + 1.1 * r(0.4142135623730951, 0.5), 0., 1.)
    for algo in ("nemo", "an05", "lu12", "lg15", "easy"):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ab.turb_ice(algo, 2.0, 10.0, Ts, tht, qs, q, W, frice=fri if algo in ("lu12", "lg15") else None, optional=(),
                        cxn=(1.5e-3, 1.3e-3, 1.4e-3) if algo == "easy" else None)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        bpc = 88 + (8 if algo == "lu12" else 0)
        print(f"{algo}: {best:.3f} ms incl. allocation of 6 outputs -> {n / best / 1e3:.0f} Mcell/s, {bpc * n / best / 1e6:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
