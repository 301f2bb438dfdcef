#!/usr/bin/env python
"""Fixture of the cells whose flux the reference itself computes in two ways (tests/golden/bistable_cells.npz) — BUILD CONTAINER ONLY
(needs /root/reference compiled by `make -C oracle refvariants`).

Round-3 soak, seed 5119 (profiles/r3_fuzz.txt item 10): ecmwf + skin, zt = zu = 10, nb_iter = 10, record 3, a dead-calm night cell:
the HIP kernel is 2.56e-10 off the reference's default build in Q_L — and equal, to every digit, to the reference built with the FMA
flag set of its own arch/ files.  The frozen metric (oracle/parity.py) rejects the value at 1.255 of its one-input ceiling 1.25.  The
rule this fixture makes testable: a value the frozen metric rejects must be ONE OF THE REFERENCE'S OWN BUILDS' values, else it is a
kernel bug.

The block: the 96 cells around the cell of the soak (same inputs, three identical records with the warm-layer state carried), through
the UNMODIFIED reference under each flag set of oracle/Makefile:
    O2     -O2 (the pinned oracle's build)            O0  arch/make.macro_ifort:11         O3  arch/make.macro_GnuLinux:17
    O3fma  -O3 -march=x86-64-v3 -ffp-contract=fast  ("-xHOST -O3" of the ifort / ifx macros on an FMA host, arch/make.macro_OCCIGEN:17)
    fast   the same plus reassociation (ifort's default -fp-model fast=1)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
from test_gpu_fuzz import _fields  # noqa: E402

ALGO, SKIN, ZT, ZU, NITER, NT = "ecmwf", True, 10.0, 10.0, 10, 3
# (seed, cell of the filtered fuzz field, record whose Q_L the metric rejected, fixture file).  Second case: round-4 soak, seed 9443 (profiles/r4_fuzz.txt
# item 10): the same configuration, a near-calm (0.2 m/s) stable night cell, record 2: 6.8e-10 off the default build in Q_L, Q_H, Evap, 6.1e-10 in tau —
# and the reference's FMA build to 1e-15 in every one of them; the kernels from before the round's last change give the same numbers.
# Third case: round-5 closing soak, seed 11029 (profiles/r5_fuzz.txt): the same configuration again, a near-calm (0.23 m/s) stable cell by day, record 3,
# Q_L 9.6e-10 off the default build (1.355 of the one-input ceiling); the kernels' arithmetic did not change in round 5.
CASES = ((5119, 1805, 3, "bistable_cells.npz"), (9443, 13104, 2, "bistable_cells_9443.npz"), (11029, 122761, 3, "bistable_cells_11029.npz"),
         # round-5 second campaign, seed 11252 (even: an unfiltered field): wind 0.33 m/s, stable, by day, record 3: Q_L 1.96e-10 off the default build = the FMA build to 1e-15
         (11252, 123576, 3, "bistable_cells_11252.npz"))
VARIANTS = ("O2", "O0", "O3", "O3fma", "fast")
IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT6 = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")


def main():
    which = [int(a) for a in sys.argv[1:]] or [c[0] for c in CASES]
    for case in CASES:
        if case[0] in which:
            generate(*case)


def generate(SEED, CELL, RECORD, FILE):
    n = 60000 + 13 * SEED
    f = _fields(SEED, n)
    if SEED % 2:                                          # odd seeds are filtered: as tests/test_gpu_fuzz.py::_fuzz_case
        keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
        f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
    lo, hi = CELL - 48, CELL + 48
    blk = {k: np.ascontiguousarray(f[k][lo:hi]) for k in IN8}
    out = {"in_" + k: blk[k] for k in IN8}
    for v in VARIANTS:
        so = po.ref_variant_so(v)
        assert os.path.exists(so), f"{so}: run `make -C oracle refvariants`"
        recs = po.run_reference(ALGO, [blk] * NT, ZT, ZU, NITER, use_skin=SKIN, variant=v)
        out["ref_" + v] = np.stack([np.stack([r[k] for k in OUT6]) for r in recs])      # [nt, 6, cells]
    # the oracle restatement and its FMA twin (what travels to the GPU box), for the record
    for tag, var in (("oracle", None), ("oracle_fma", "fma")):
        s = po.OracleSession(ALGO, hi - lo, NT, SKIN, variant=var)
        rows = []
        for jt in range(1, NT + 1):
            o = s.compute(jt, ZT, ZU, NITER, *[blk[k] for k in IN8[:6]], rad_sw=blk["rad_sw"], rad_lw=blk["rad_lw"])
            rows.append(np.stack([o[k] for k in OUT6]))
        out["ref_" + tag] = np.stack(rows)
    out["meta"] = np.array(f"seed {SEED} {ALGO} skin zt={ZT} zu={ZU} nb_iter={NITER} nt={NT} record {RECORD}; cells {lo}..{hi - 1} of the filtered fuzz field, "
                           f"the soak's cell is index {CELL - lo}; variants {','.join(VARIANTS)}; planes {','.join(OUT6)}")
    out["cell"] = np.array(CELL - lo)
    out["record"] = np.array(RECORD)
    path = os.path.join(ROOT, "tests", "golden", FILE)
    np.savez_compressed(path, **out)
    c = CELL - lo
    base = out["ref_O2"]
    print("wrote", path)
    for v in VARIANTS[1:] + ("oracle", "oracle_fma"):
        d = out["ref_" + v] - base
        print(f"{v:10s}: cell {CELL} record {RECORD}  dQL {d[RECORD - 1, 0, c]:+.4e}  dQH {d[RECORD - 1, 1, c]:+.4e}   other cells: max |d|/|ref| "
              f"{np.max(np.abs(np.delete(d, c, axis=2)) / np.maximum(np.abs(np.delete(base, c, axis=2)), 1e-30)):.2e}")


if __name__ == "__main__":
    main()
