#!/usr/bin/env python
"""Conditioning study of the cells where HIP and the oracle differ by more than 1e-10 (build container: needs /root/reference).

Input: the dump of tools/outlier_dump.py (GPU box): every cell of the 4320x3600 benchmark grid and of the wide fuzz fields whose HIP
flux differs from the oracle's by more than 1e-10 relative (floor 1e-6 of the field maximum), with inputs, oracle and HIP values.
Each such cell is put through
  * the UNMODIFIED reference compiled with the flag sets of its own arch/ files (oracle/Makefile `refvariants`):
    -O0, -O2 (the pinned build), -O3, "-xHOST -O3" on an FMA host (contraction), the same with reassociation;
  * the pinned reference build with every input moved by +-1 ulp (and +-2, +-4, +-8 ulp) in turn;
  * the C oracle with every input moved by +-1 ulp in turn, and the oracle compiled with FMA contraction.
"Reference spread" of a value: S_ref = max |x - x_O2| over the reference builds and the +-1 ulp runs (S_ref4 / S_ref8: the runs with
moves of up to 4 / 8 ulp as well):
what the reference itself leaves undefined.  Output: tests/golden/illcond_cells.npz (inputs, reference O2 values, spreads, the HIP
error measured when the fixture was made) and a text report (profiles/r2_illcond_study.txt).  The GPU tests then demand of every
fixture value |HIP - ref_O2| <= max(1e-10 bar, S_ref8) — the backward clause of oracle/parity.py, evaluated on the reference itself —
and of every other cell they meet that rule with the oracle's response.

    python tools/illcond_study.py gpurun_out/outliers_r2.npz [--write]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT6 = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
REF_VARIANTS = ("O0", "O3", "O3fma", "fast")
# field scales of the configuration the cells came from (the floors of the parity metric are fractions of the field maximum):
# taken from the dump's own reference values would be wrong (only outliers are in it), so they are recomputed on the fly


def reference_records(algo, skin, zt, zu, niter, nt, f, variant=None):
    recs = [{k: f[k] for k in IN8 if skin or k not in ("rad_sw", "rad_lw")} for _ in range(nt)]
    res = po.run_reference(algo, recs, zt, zu, niter, use_skin=skin, variant=variant)
    return np.stack([np.stack([r[k] if (skin or k != "t_s") else f["sst"] for k in OUT6]) for r in res])    # [nt, 6, n]


def oracle_records(algo, skin, zt, zu, niter, nt, f, variant=None):
    s = po.OracleSession(algo, f["sst"].size, nt, skin, variant=variant)
    out = []
    for jt in range(1, nt + 1):
        o = s.compute(jt, zt, zu, niter, *[f[k] for k in IN8[:6]], rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
        out.append(np.stack([o[k] for k in OUT6]))
    return np.stack(out)


def field_scales(meta):
    """max |field| of the configuration's full input set (oracle, a 1/16 subsample is plenty for a maximum to 3 digits)."""
    algo, skin, zt, zu, niter = meta["algo"], meta["skin"], meta["zt"], meta["zu"], meta["niter"]
    if "grid" in meta:
        f = po.synth_fields(meta["grid"][0], meta["grid"][1] // 16)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_fuzz import _fields
        f = _fields(meta["seeds"][0], meta["seeds"][2])
        keep = np.hypot(f["u_zu"], f["v_zu"]) < 30.0
        f = {k: np.ascontiguousarray(v[keep]) for k, v in f.items()}
    o = oracle_records(algo, skin, zt, zu, niter, 1, f)
    return np.abs(o[0]).max(axis=1)


def main():
    src = sys.argv[1]
    write = "--write" in sys.argv
    d = np.load(src)
    meta = json.loads(str(d["meta"]))
    fixture, lines = {}, []
    fmeta = {}
    for tag, m in meta.items():
        if tag + "_inputs" not in d:
            continue
        algo, skin, zt, zu, niter, nt = m["algo"], m["skin"], m["zt"], m["zu"], m["niter"], m["nt"]
        inp = d[tag + "_inputs"]
        n = inp.shape[1]
        f = {k: np.ascontiguousarray(inp[i]) for i, k in enumerate(IN8)}
        got, ref_dump = d[tag + "_got"], d[tag + "_ref"]
        nf = 6 if skin else 5
        base = reference_records(algo, skin, zt, zu, niter, nt, f)
        orc = oracle_records(algo, skin, zt, zu, niter, nt, f)
        assert np.allclose(orc[:, :nf], ref_dump[:, :nf], rtol=0, atol=0), "oracle here != oracle on the GPU box"
        scale = field_scales(m)[:nf]
        top = scale[None, :, None]
        # the reference's own spread
        spread_ref = np.zeros_like(base)
        per_var = {}
        for v in REF_VARIANTS:
            x = reference_records(algo, skin, zt, zu, niter, nt, f, variant=v)
            per_var[v] = np.abs(x - base)
            spread_ref = np.maximum(spread_ref, per_var[v])
        # the sensitivity to +-1 ulp on each input in turn: of the reference (pinned -O2 build) and of the oracle (+ its FMA build)
        spread_orc = np.abs(oracle_records(algo, skin, zt, zu, niter, nt, f, variant="fma") - orc)
        spread_ulp = np.zeros_like(base)
        spread_ulp4 = np.zeros_like(base)
        spread_ulp8 = np.zeros_like(base)
        for k in IN8[: (8 if skin else 6)]:
            for sgn in (+1, -1):
                g = dict(f)
                g[k] = np.nextafter(f[k], sgn * np.inf)
                spread_orc = np.maximum(spread_orc, np.abs(oracle_records(algo, skin, zt, zu, niter, nt, g) - orc))
                spread_ulp = np.maximum(spread_ulp, np.abs(reference_records(algo, skin, zt, zu, niter, nt, g) - base))
                for ulps in (2, 4, 8):
                    gk = dict(f)
                    gk[k] = f[k] + sgn * ulps * np.abs(np.nextafter(f[k], np.inf) - f[k])
                    mv = np.abs(reference_records(algo, skin, zt, zu, niter, nt, gk) - base)
                    if ulps <= 4:
                        spread_ulp4 = np.maximum(spread_ulp4, mv)
                    spread_ulp8 = np.maximum(spread_ulp8, mv)
        spread_builds = spread_ref
        spread_ref = np.maximum(spread_builds, spread_ulp)       # S of oracle/parity.py, measured on the reference itself
        spread_ref4 = np.maximum(spread_ref, spread_ulp4)
        spread_ref8 = np.maximum(spread_ref, spread_ulp8)
        err = np.abs(got - base)[:, :nf]
        bar6 = 1e-10 * np.maximum(np.abs(base[:, :nf]), 1e-6 * top)
        bar4 = 1e-10 * np.maximum(np.abs(base[:, :nf]), 1e-4 * top)
        beyond6 = err > bar6
        beyond4 = err > bar4
        sr, so = spread_ref[:, :nf], spread_orc[:, :nf]
        nb6, nb4 = int(beyond6.sum()), int(beyond4.sum())
        cov_b = int((beyond6 & (spread_builds[:, :nf] >= err)).sum())
        cov_ref = int((beyond6 & (sr >= err)).sum())
        cov_ref4 = int((beyond6 & (spread_ref4[:, :nf] >= err)).sum())
        cov_ref8 = int((beyond6 & (spread_ref8[:, :nf] >= err)).sum())
        ratio = err[beyond6] / np.maximum(sr[beyond6], 1e-300)
        ratio_o = err[beyond6] / np.maximum(so[beyond6], 1e-300)
        agree = so[beyond6] / np.maximum(sr[beyond6], 1e-300)
        lines.append(f"{tag} {algo} skin={int(skin)} zt={zt} zu={zu} n={niter} records={nt}: {n} flagged cells; values beyond 1e-10 with floor 1e-6: {nb6} "
                     f"(floor 1e-4: {nb4}); HIP error <= spread of the reference builds alone on {cov_b}, <= reference spread S_ref (builds + inputs +-1 ulp) on "
                     f"{cov_ref}, <= S_ref4 (inputs within +-4 ulp) on {cov_ref4}, <= S_ref8 (within +-8 ulp) on {cov_ref8} of {nb6}; HIP error / S_ref: median {np.median(ratio):.2f}, p90 {np.quantile(ratio, 0.9):.2f}, "
                     f"max {ratio.max():.2f}; HIP error / S_oracle max {ratio_o.max():.2f}; S_oracle / S_ref in [{agree.min():.2f}, {agree.max():.2f}]; "
                     f"largest HIP error {float((err / bar6).max()):.1f} bars; largest move of a reference build, in bars: "
                     + ", ".join(f"{v} {float((per_var[v][:, :nf] / bar6).max()):.1f}" for v in REF_VARIANTS))
        print(lines[-1], flush=True)
        fixture[tag + "_inputs"] = inp
        fixture[tag + "_ref"] = base
        fixture[tag + "_spread_ref"] = spread_ref
        fixture[tag + "_spread_ref_ulp"] = spread_ulp            # the +-1 ulp part alone (pinned build)
        fixture[tag + "_spread_ref4"] = spread_ref4
        fixture[tag + "_spread_ref8"] = spread_ref8
        fixture[tag + "_spread_oracle"] = spread_orc
        fixture[tag + "_hip_err_when_made"] = np.abs(got - base)
        fixture[tag + "_scale"] = scale
        fmeta[tag] = m
    if write:
        fixture["meta"] = np.array(json.dumps(fmeta))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "illcond_cells.npz"), **fixture)
        with open(os.path.join(ROOT, "profiles", "r2_illcond_study.txt"), "w") as fh:
            fh.write(__doc__.split("\n\n")[0] + "\n\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
