"""oracle/parity.py — TEST INFRASTRUCTURE ONLY: the parity metric of the hot path (tests/, tools/, __graft_entry__.smoke()).

north_star: fluxes within 1e-10 relative of the Fortran reference (fp64).  SURVEY §8d: relative error with a floor of 1e-6 of the
field maximum.  Measured on all 15 552 000 cells of the benchmark grid (tools/outlier_dump.py, profiles/r2_illcond_study.txt): 200-250
values per field of every algorithm lie beyond that bar, all of them fluxes that vanish by cancellation (theta_zu - T_s ~ 1e-5 K,
q_zu - q_s ~ 1e-9): there the REFERENCE ITSELF answers a one-ulp change of one input, or a rebuild with the "-xHOST -O3" of its own
arch/ files (FMA contraction), by more than the bar (tools/illcond_study.py, tests/golden/illcond_cells.npz).  A forward error of
1e-10 is not defined for such a cell; what is defined is the backward error.  The metric is therefore, per cell and field:

    |got - ref| <= 1e-10 * max(|ref|, 1e-6 * max|ref|)                        (forward clause)
 or |got - ref| <= S,   S = sum over the inputs of the cell of the largest change of the oracle's value when THAT input alone moves
                            by up to BACKWARD_ULPS = 8 ulp (both directions, moves of 1, 2, 4 and 8 ulp), or the change under FMA
                            contraction if that is larger
                            (backward clause, the componentwise backward error of numerical analysis to first order: got is what
                            the reference computes for inputs that are EACH within 8 ulp, 1.8e-15 relative, of the given ones)

and the number of values that need the second clause is budgeted (illcond_allowance(): ILLCOND_BUDGET of the values, at least one per 512 up to 4 on small fields) so that it cannot become
a blanket excuse; the error may not exceed ONE_INPUT_CEILING times the largest response to ONE input either (below).
The counts with the round-1 floor (1e-4) are reported next to those with the 1e-6 floor, and the error in units of the ONE-ulp
response next to the verdict.

Where the 8 comes from (tools/fuzz_probe.py, profiles/r2_fuzz_wide.txt).  The clause was first written as |got - ref| <= 4 S1 (S1:
one-ulp moves), calibrated on the 4 990 flagged values of the conditioning study (largest error 3.0 S1).  A campaign of 120 further
seeds (7e8 values, 2 400 of them beyond the forward bar) found the tail: 99.3 % of the flagged values within 4 S1, one cell at 4.5
and one at 5.7 (latent-heat fluxes of 7e-3 and 6e-2 W/m2, errors of 6e-12 and 5e-11 W/m2); their response is linear in the size of
the move (error / response to moves of <= 1, 2, 3, 4, 8 ulp: 5.7, 2.6, 1.8, 1.5, 0.77), so the clause is stated in the moves
themselves, with a bound the tail stays under by a margin.
One input or all of them.  Until the last campaign of round 2 S was the largest response to a move of ONE input (a sufficient
condition, stricter than the definition).  After 4 700 more tests (1.6e9 values, profiles/r2_fuzz_wide.txt) one value of an ECMWF +
skin cell iterating on its clamps — where the response is a jump, the same for one ulp as for eight — stood at 1.10 of it (0.42 of
the response to all inputs moving together).  The clause is now the definition itself: every input may move, S is the sum of the
single-input responses; the ratio against the one-input S is still reported (backward_ratio_one_input_max).
"""
from __future__ import annotations

import json

import numpy as np

TOL_REL = 1e-10
FLOOR_FRAC = 1e-6
FLOOR_FRAC_R1 = 1e-4       # the floor round 1 asserted; still counted and reported
BACKWARD_ULPS = 8             # largest single-input move of the backward clause
ULP_MOVES = (1, 2, 4, 8)      # the moves sampled
ILLCOND_BUDGET = 2e-4      # largest tolerated share of values that pass by the backward clause only (measured: <= 5e-5)
ILLCOND_MIN_COUNT = 1      # ... but at least this many values of a field, one more per ILLCOND_SMALL_N values up to ILLCOND_MAX_SMALL
ILLCOND_SMALL_N = 512      #     (round 2 allowed 4 whatever the size: four of a 64-cell test could skip the forward clause; the golden
ILLCOND_MAX_SMALL = 4      #     sweeps of 2 048 cells, which cross theta_zu = T_s on purpose, hold up to 3 such values per field)
ONE_INPUT_CEILING = 1.25   # largest tolerated error / largest response to a move of ONE input (the stricter, sufficient form of the
#                            clause; measured over 1.9e9 values: 1.10, profiles/r2_fuzz_wide.txt entry 15): asserted since round 3
# FROZEN (round 3): tests/test_parity_metric.py pins every constant above and the formula of S below, and shows with injected errors
# that the metric fails.  A new failure is a kernel bug until the reference rebuilt under its own flags shows otherwise.
IN8 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT6 = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")


def rel_err(a, b, floor_frac=FLOOR_FRAC, scale=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    top = float(np.max(np.abs(b))) if scale is None else float(scale)
    return np.abs(a - b) / np.maximum(np.abs(b), floor_frac * max(top, 1e-300))


def _move(x, sgn, ulps):
    for _ in range(ulps):
        x = np.nextafter(x, sgn * np.inf)
    return x


class OracleSensitivity:
    """S of the backward clause for the cells of one configuration.  `records`: list (one per time record) of dicts of flat
    float64 input arrays (keys IN8; rad_* only with skin); a single dict = the same inputs for every record."""

    def __init__(self, po, algo, skin, zt, zu, niter, records, nt=1, hum_type="sh", isecday_utc=12, lon=None):
        self.po, self.algo, self.skin, self.zt, self.zu, self.niter, self.nt = po, algo, bool(skin), zt, zu, niter, nt
        self.records = [records] * nt if isinstance(records, dict) else list(records)
        self.hum_type, self.isecday, self.lon = hum_type, isecday_utc, lon
        self._cache = {}

    def _run(self, recs, idx, variant=None):
        s = self.po.OracleSession(self.algo, idx.size, self.nt, self.skin, self.hum_type, variant=variant)
        lon = None if self.lon is None else np.ascontiguousarray(np.asarray(self.lon)[idx])
        out = []
        for jt, r in enumerate(recs, 1):
            isd = self.isecday[jt - 1] if np.ndim(self.isecday) else self.isecday
            o = s.compute(jt, self.zt, self.zu, self.niter, *[r[k] for k in IN8[:6]], rad_sw=r.get("rad_sw") if self.skin else None,
                          rad_lw=r.get("rad_lw") if self.skin else None, isecday_utc=int(isd), lon=lon)
            out.append(np.stack([o[k] for k in OUT6]))
        return np.stack(out)          # [nt, 6, m]

    def __call__(self, jt, idx):
        """(S, S1): dicts key -> array[len(idx)] for record jt (1-based): response to moves of up to BACKWARD_ULPS, to one-ulp moves."""
        idx = np.asarray(idx)
        key = idx.tobytes()
        if key not in self._cache:
            recs = [{k: np.ascontiguousarray(np.asarray(r[k], dtype=np.float64)[idx]) for k in IN8 if r.get(k) is not None}
                    for r in self.records]
            base = self._run(recs, idx)
            Sfma = np.abs(self._run(recs, idx, variant="fma") - base)
            S1 = None
            per = {}                       # input -> largest response to a move of that input alone, up to BACKWARD_ULPS
            for ulps in ULP_MOVES:
                for k in IN8[:8 if self.skin else 6]:
                    for sgn in (1.0, -1.0):
                        pert = [dict(r, **{k: _move(r[k], sgn, ulps)}) for r in recs]
                        d = np.abs(self._run(pert, idx) - base)
                        per[k] = d if k not in per else np.maximum(per[k], d)
                if S1 is None:
                    S1 = np.maximum.reduce([Sfma] + list(per.values()))   # FMA contraction and one-ulp moves: reported, not asserted
            Sone = np.maximum.reduce([Sfma] + list(per.values()))         # one input at a time
            S = np.maximum(Sfma, np.sum(list(per.values()), axis=0))      # all inputs at once, to first order
            self._cache[key] = (S, S1, Sone)
        S, S1, Sone = self._cache[key]
        self.last_one_input = {k: Sone[jt - 1, i] for i, k in enumerate(OUT6)}
        return {k: S[jt - 1, i] for i, k in enumerate(OUT6)}, {k: S1[jt - 1, i] for i, k in enumerate(OUT6)}


def parity_report(got, ref, keys, tol=TOL_REL, sens=None, jt=1, scales=None):
    """Per field: forward-clause figures with both floors, and (if `sens`) what the backward clause says about the values beyond
    the 1e-6-floored bar.  `scales`: field maxima to use instead of max|ref| (subsets of a larger field)."""
    rep = {}
    for k in keys:
        g, r = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        top = float(np.max(np.abs(r))) if not scales else float(scales[k])
        err = np.abs(g - r)
        e6 = err / np.maximum(np.abs(r), FLOOR_FRAC * max(top, 1e-300))
        e4 = err / np.maximum(np.abs(r), FLOOR_FRAC_R1 * max(top, 1e-300))
        bad = np.nonzero(~(e6 <= tol))[0]
        row = dict(n=int(r.size), max_rel=float(np.max(e6)), p9999=float(np.quantile(e6, 0.9999)), n_gt_tol=int(bad.size),
                   max_rel_floor4=float(np.max(e4)), n_gt_tol_floor4=int((~(e4 <= tol)).sum()),
                   max_abs_over_scale=float(np.max(err)) / max(top, 1e-300), n_nonfinite=int((~np.isfinite(g)).sum()))
        if bad.size and sens is not None and row["n_nonfinite"] == 0:
            S, S1 = sens(jt, bad)
            ratio = err[bad] / np.maximum(np.asarray(S[k]), 1e-300)
            row["backward_ratio_max"] = float(np.max(ratio))                # <= 1: inside the response to moves of <= 8 ulp of all inputs
            one = getattr(sens, "last_one_input", None)
            if one is not None:                                             # the same against moves of ONE input at a time (reported)
                row["backward_ratio_one_input_max"] = float(np.max(err[bad] / np.maximum(np.asarray(one[k]), 1e-300)))
            row["backward_ulps_max"] = float(np.max(err[bad] / np.maximum(np.asarray(S1[k]), 1e-300)))   # in units of the one-ulp response
            row["n_unexplained"] = int((ratio > 1.0).sum())
        elif bad.size:
            row["n_unexplained"] = int(bad.size)
        else:
            row["n_unexplained"] = 0
        rep[k] = row
    return rep


def illcond_allowance(n, budget=ILLCOND_BUDGET):
    """Largest number of values of an n-value field that may pass by the backward clause only."""
    return max(int(budget * n), min(ILLCOND_MAX_SMALL, max(ILLCOND_MIN_COUNT, n // ILLCOND_SMALL_N)))


def check_parity(got, ref, keys, tol=TOL_REL, sens=None, jt=1, label="", budget=ILLCOND_BUDGET, scales=None, quiet=False):
    rep = parity_report(got, ref, keys, tol, sens, jt, scales)
    if not quiet:
        print(label, json.dumps(rep))
    for k in keys:
        r = rep[k]
        assert r["n_nonfinite"] == 0, (label, k, r)
        assert r["n_unexplained"] == 0, (label, k, r)          # every value: forward clause, or backward clause (moves <= 8 ulp)
        assert r["n_gt_tol"] <= illcond_allowance(r["n"], budget), (label, k, r)   # ... and only a handful may need the latter
        # ... none of them far beyond the stricter one-input form of the clause (so that form cannot regress silently)
        assert r.get("backward_ratio_one_input_max", 0.0) <= ONE_INPUT_CEILING, (label, k, r)
    return rep
