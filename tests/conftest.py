import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# ---- parity bar -------------------------------------------------------------------------------
# north_star: fluxes within 1e-10 relative of the Fortran reference (fp64).
# HOT PATH (aerobulk_compute: tests/test_gpu_parity, _golden, _fuzz, _fullsize, _hosts, _illcond): the metric of oracle/parity.py,
#   |got-ref| <= 1e-10 max(|ref|, 1e-6 max|ref|)      or   backward error <= 8 ulp of every input (the frozen clause of oracle/parity.py, pinned by tests/test_parity_metric.py), budgeted;
#   see that module and profiles/r2_illcond_study.txt for the reference-side evidence.  -> assert_hot_parity()
# NEXT-TIER ROWS (diagnostics, TURB_* series, sea ice) keep the round-1 form with their own stated tolerances:
#   |got-ref| <= TOL_REL * max(|ref|, FLOOR_FRAC*max|ref|) and |got-ref| <= TOL_ABS_FRAC*max|ref| for every cell.  -> assert_parity()
TOL_REL = 1e-10
FLOOR_FRAC = 1e-4
TOL_ABS_FRAC = 1e-12


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.hookimpl(trylast=True)
def pytest_sessionstart(session):
    global _TR
    _TR = session.config.pluginmanager.get_plugin("terminalreporter")


def rel_err(a, b, floor_frac=FLOOR_FRAC):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), floor_frac * max(float(np.max(np.abs(b))), 1e-300))
    return np.abs(a - b) / scale


def parity_report(got, ref, keys, tol=TOL_REL):
    rep = {}
    for k in keys:
        g, r = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        e = rel_err(g, r)
        e6 = rel_err(g, r, 1e-6)
        amax = float(np.max(np.abs(r)))
        rep[k] = dict(max_rel=float(e.max()), p9999=float(np.quantile(e, 0.9999)), n_bad=int((e > tol).sum()),
                      n_gt_tol_floor6=int((e6 > tol).sum()), max_abs_over_scale=float(np.max(np.abs(g - r)) / max(amax, 1e-300)),
                      n_nonfinite=int((~np.isfinite(g)).sum()))
    return rep


def assert_parity(got, ref, keys, tol=TOL_REL, abs_frac=TOL_ABS_FRAC, label=""):
    rep = parity_report(got, ref, keys, tol)
    print(label, json.dumps(rep))
    for k in keys:
        assert rep[k]["n_nonfinite"] == 0, (label, k, rep[k])
        assert rep[k]["n_bad"] == 0, (label, k, rep[k])
        assert rep[k]["max_abs_over_scale"] <= abs_frac, (label, k, rep[k])
    return rep


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


def load_golden_case(case):
    inp = dict(np.load(os.path.join(GOLDEN, "sweep_inputs.npz")))
    out = dict(np.load(os.path.join(GOLDEN, case["name"] + ".npz")))
    if "hum_zt" in out:
        inp["hum_zt"] = out.pop("hum_zt")
    keys = ("ql", "qh", "tau_x", "tau_y", "evap") + (("t_s",) if case["skin"] else ())
    recs = [{k: out[f"jt{jt}_{k}"] for k in keys} for jt in range(1, case["nt"] + 1)]
    return inp, recs, keys


def assert_hot_parity(got, ref, keys, sens=None, jt=1, label="", **kw):
    """Hot-path parity (oracle/parity.py): forward clause with the 1e-6 floor, else backward clause through `sens`
    (an oracle.parity.OracleSensitivity of the same configuration and inputs)."""
    from oracle import parity
    return parity.check_parity(got, ref, keys, sens=sens, jt=jt, label=label, **kw)


def sensitivity(po, algo, skin, zt, zu, niter, fields, nt=1, **kw):
    """OracleSensitivity for fields given with either spelling of the wind names (u_zu / U_zu)."""
    from oracle import parity
    def norm(f):
        g = {k: f[k] for k in ("sst", "t_zt", "hum_zt", "slp") if k in f}
        g["u_zu"] = f["u_zu"] if "u_zu" in f else f["U_zu"]
        g["v_zu"] = f["v_zu"] if "v_zu" in f else f["V_zu"]
        for k in ("rad_sw", "rad_lw"):
            if f.get(k) is not None:
                g[k] = f[k]
        return {k: np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()) for k, v in g.items()}
    recs = norm(fields) if isinstance(fields, dict) else [norm(f) for f in fields]
    return parity.OracleSensitivity(po, algo, skin, zt, zu, niter, recs, nt=nt, **kw)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    if not os.path.exists(pyoracle.ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so"),
                               os.path.join(ROOT, "oracle", "liboracle_fma.so")])
    return pyoracle


# ---- run log: the tail of pytest's output names the test that is running and what each one cost ------------------------------
import time as _time

_T0 = _time.time()


_TR = None


def _log(msg):
    if _TR is not None:            # through pytest's own terminal writer: lands in whatever captures pytest's stdout
        _TR.ensure_newline()
        _TR.write_line(msg)
    else:
        sys.stderr.write(msg + "\n")
        sys.stderr.flush()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    t = _time.time()
    _log(f"[ab-test] START {item.nodeid}  t+{t - _T0:7.1f}s")
    yield
    _log(f"[ab-test] END   {item.nodeid}  {(_time.time() - t):7.2f}s  t+{_time.time() - _T0:7.1f}s")


# ---- order and budget of the GPU suite (round-3 verdict: the driver's 1 200 s limit cut the run before the golden vectors) ------------
# Tier 1: the hot path against the reference's own vectors and the C-ABI hosts — what the parity claim rests on.  Tier 2: the "next"
# rows and the properties.  Tier 3: soak (more seeds, more full-size configurations, adversarial fields, the other bench launches):
# runs last and only while the session is inside its wall budget (AB_TEST_BUDGET_S, default 600 s; 0 = no budget); tier 1 / 2 never skip.
BUDGET_S = float(os.environ.get("AB_TEST_BUDGET_S", "600"))
_FULLSIZE_T1 = ("coare3p6-True-5", "coare3p6-False-8")                    # BASELINE configs 3 and 2 (kernel) on every cell of the ORCA12 grid
_BENCH_T2 = ("[2-extra0]", "[2-extra6]", "[2-extra8]")                     # default, config 4, config 5: the other torchrun launches are tier 3


def tier_of(nodeid):
    f = nodeid.split("::")[0].rsplit("/", 1)[-1]
    name = nodeid.split("::", 1)[1] if "::" in nodeid else ""
    if f in ("test_gpu_golden.py", "test_gpu_parity.py", "test_illcond_cells.py", "test_bistable_cells.py", "test_gpu_cu_kernel.py", "test_phymbl.py", "test_reference_drivers.py", "test_skin_modules.py"):
        return 1                                                           # (test_phymbl.py: the Fortran side of the drop-in boundary, incl. the reference's unchanged example)
    if f == "test_gpu_hosts.py":
        if name.startswith("test_bench_sharded_path_several_ranks_one_gpu"):
            return 2 if any(t in name for t in _BENCH_T2) else 3
        return 1
    if f == "test_gpu_fullsize.py":
        return 1 if any(t in name for t in _FULLSIZE_T1) else 3
    if f == "test_gpu_mixed.py":
        if "orca36" in name:
            return 1.5        # config 5 at its full size: tier 1, but behind the rest of it (665 s on one lease, 5-15 s on the others; it logs its stages)
        return 1 if "test_mixed_sessions_meet_the_restated_tolerance" in name else 2
    if f == "test_gpu_sharded.py":
        return 1 if ("test_device_resident_shards_and_gather[coare3p6-True-3" in name or "rccl" in name) else 2
    if f == "test_gpu_fuzz.py":
        if "test_both_humidity_differences" in name:
            return 2
        seed = name[name.index("[") + 1:].split("-")[-1].rstrip("]") if "[" in name else ""
        return 2 if seed == "100" else 3                                   # the first seed of every configuration; the others soak
    if f == "test_gpu_adversarial.py":
        return 3
    return 2


def pytest_collection_modifyitems(config, items):
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: (tier_of(it.nodeid) if it.get_closest_marker("gpu") else 0, order[id(it)]))


def pytest_runtest_setup(item):
    if BUDGET_S > 0 and item.get_closest_marker("gpu") and tier_of(item.nodeid) == 3 and _time.time() - _T0 > BUDGET_S:
        pytest.skip(f"time budget: soak tier (3) not started after {BUDGET_S:.0f} s of session (AB_TEST_BUDGET_S=0 lifts it)")


# ---- the oracle on the host cores: fresh interpreters, never a fork of the process that holds the HIP runtime -------------------------
def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup quota (the driver's boxes are shared)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 48))


_POOL = None


def oracle_pool():
    """Process pool for oracle blocks.  forkserver: the workers descend from a server that is a fresh interpreter (started here at
    conftest import, before any test touches the GPU), not from the pytest process with its HIP / torch threads and locks."""
    global _POOL
    if _POOL is None:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["numpy"])
        _POOL = ProcessPoolExecutor(usable_cpus(), mp_context=ctx)
    return _POOL


_TIER_COUNTS = {}


def pytest_runtest_logreport(report):
    # what ran in which tier: a green exit with the soak tier skipped by the budget must not read like a full run (one line at the end)
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        if "gpu" in getattr(report, "keywords", {}):
            t = int(tier_of(report.nodeid))
            c = _TIER_COUNTS.setdefault(t, {"passed": 0, "failed": 0, "skipped": 0})
            c[report.outcome] = c.get(report.outcome, 0) + 1


def pytest_terminal_summary(terminalreporter):
    if _TIER_COUNTS:
        parts = [f"tier {t}: {c['passed']} passed, {c['failed']} failed, {c['skipped']} skipped" for t, c in sorted(_TIER_COUNTS.items())]
        terminalreporter.write_line("[ab-test] GPU tiers — " + "; ".join(parts) + (f" (soak budget {BUDGET_S:.0f} s)" if BUDGET_S > 0 else " (no budget)"))


def pytest_sessionfinish(session, exitstatus):
    global _POOL
    if _POOL is not None:
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None


from oracle_workers import IN8_, OUT6_, _oracle_cells, _oracle_rows  # noqa: E402  (a module of its own: the workers import it, not conftest)


_FULL_CACHE = {}


def oracle_full_grid(algo, skin, niter, ni, nj):
    """The oracle on every cell of the synthetic ni x nj grid, in j-blocks on the host cores; computed once per session and
    configuration (tests that share one reuse it)."""
    key = (algo, bool(skin), niter, ni, nj)
    if key not in _FULL_CACHE:
        nproc = usable_cpus()
        per = max(1, -(-nj // (nproc * 4)))
        ref = {}
        jobs = [(ROOT, algo, bool(skin), niter, ni, nj, j0, min(per, nj - j0)) for j0 in range(0, nj, per)]
        for j0, o in oracle_pool().map(_oracle_rows, jobs):
            for k, v in o.items():
                ref.setdefault(k, np.empty(ni * nj))[j0 * ni:j0 * ni + v.size] = v
        if len(_FULL_CACHE) >= 2:                  # 15.5 M cells x 6 fields = 750 MB per entry
            _FULL_CACHE.pop(next(iter(_FULL_CACHE)))
        _FULL_CACHE[key] = ref
    return _FULL_CACHE[key]


def oracle_on_cells(algo, skin, niter, f64, nt=1):
    """The oracle on the cells of `f64` (dict of flat fp64 arrays, keys IN8_), cut into chunks over the host cores: list over records."""
    n = f64["sst"].size
    nproc = usable_cpus()
    edges = np.linspace(0, n, max(1, min(nproc * 4, n // 1000 or 1)) + 1).astype(np.int64)
    jobs = [(ROOT, algo, bool(skin), niter, nt, [np.ascontiguousarray(f64[k][a:b]) for k in IN8_]) for a, b in zip(edges[:-1], edges[1:])]
    parts = list(oracle_pool().map(_oracle_cells, jobs))
    return [{k: np.concatenate([p[jt][k] for p in parts]) for k in parts[0][jt]} for jt in range(nt)]


if os.environ.get("AB_NO_ORACLE_POOL") != "1":
    try:                                           # start the fork server now: conftest is imported before any test initialises HIP
        import multiprocessing.forkserver as _fs
        import multiprocessing as _mp
        _mp.get_context("forkserver").set_forkserver_preload(["numpy"])
        _fs.ensure_running()
    except Exception as _e:                        # (oracle_pool() will raise the real error when a test needs it)
        _log(f"[ab-test] fork server not started at import: {_e}")
