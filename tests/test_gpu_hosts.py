"""GPU: the host sides above the C ABI — Fortran module (mod_aerobulk, ISO_C_BINDING), C++ aerobulk::model,
Python mirror aerobulk_model — all drive the same HIP kernels.  Checked against the reference's captured
example output (doc/ex_ab.dat -> tests/golden/ex_ab.json) and against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_hot_parity, sensitivity

pytestmark = pytest.mark.gpu
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
COLS = ("qh", "ql", "evap", "t_s", "tau_x", "tau_y")


def _parse_results(text):
    res = {}
    for line in text.splitlines():
        if line.startswith("RESULT"):
            p = line.split()
            v = np.array([float(x) for x in p[2:14]]).reshape(6, 2)
            res[p[1]] = dict(zip(COLS, v))
    return res


def _check_against_ex_ab(res):
    ex = json.load(open(os.path.join(GOLDEN, "ex_ab.json")))
    assert set(res) == set(ex["cases"])
    for algo, c in ex["cases"].items():
        tol = 1e-3 if c.get("loose") else 6e-7
        np.testing.assert_allclose(res[algo]["qh"], c["qh"], rtol=tol)
        np.testing.assert_allclose(res[algo]["ql"], c["ql"], rtol=tol)
        np.testing.assert_allclose(res[algo]["evap"] * 86400.0, c["evap_mm_day"], rtol=tol)
        np.testing.assert_allclose(res[algo]["tau_x"], c["tau_x"], rtol=tol)
        np.testing.assert_array_equal(res[algo]["tau_y"], 0.0)
        if c["skin"]:
            np.testing.assert_allclose(res[algo]["t_s"] - 273.15, c["t_s_degC"], rtol=tol)


def _check_against_oracle(res, oracle, niter):
    ex = json.load(open(os.path.join(GOLDEN, "ex_ab.json")))
    i = ex["inputs"]
    f = {k: np.array(i[k], dtype=np.float64) for k in IN6 + ("rad_sw", "rad_lw")}
    for algo, c in ex["cases"].items():
        o = oracle.OracleSession(algo, 2, 1, c["skin"]).compute(1, 2.0, 10.0, niter, *[f[k] for k in IN6],
                                                                rad_sw=f["rad_sw"] if c["skin"] else None,
                                                                rad_lw=f["rad_lw"] if c["skin"] else None)
        keys = ("ql", "qh", "tau_x", "evap") + (("t_s",) if c["skin"] else ())
        for k in keys:
            np.testing.assert_allclose(res[algo][k], o[k], rtol=1e-10, err_msg=f"{algo} {k}")


def test_fortran_host_example_driver(oracle):
    exe = os.path.join(ROOT, "aerobulk_amd", "fortran", "example_call_aerobulk.x")
    if not os.path.exists(exe):
        pytest.skip("Fortran host not built (amdflang absent)")
    out = subprocess.run([exe, "50"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "----- AeroBulk_init -----" in out.stdout and "----- AeroBulk_bye -----" in out.stdout
    assert "will use the Cool-skin & Warm-layer scheme of `coare3p6`" in out.stdout
    _check_against_ex_ab(_parse_results(out.stdout))
    out8 = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=300)
    _check_against_oracle(_parse_results(out8.stdout), oracle, 8)


def test_cxx_host_example_driver(oracle):
    exe = os.path.join(ROOT, "aerobulk_amd", "csrc", "example_call_aerobulk_cxx.x")
    if not os.path.exists(exe):
        pytest.skip("C++ example not built")
    out = subprocess.run([exe, "50"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    _check_against_ex_ab(_parse_results(out.stdout))
    out8 = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=300)
    _check_against_oracle(_parse_results(out8.stdout), oracle, 8)


def test_python_aerobulk_model_protocol(oracle):
    """AEROBULK_MODEL protocol through the process-global session: INIT at jt==1, sticky Niter, T_s = sst when
    radiation is given without l_use_skin (mod_aerobulk_compute.f90:132,206), 2-D (Ni,Nj) arrays."""
    import aerobulk_amd as ab
    ni, nj = 64, 48
    f = oracle.synth_fields(ni, nj)
    F = {k: v.reshape((ni, nj), order="F") for k, v in f.items()}
    args = [F[k] for k in IN6]
    r1 = ab.aerobulk_model(1, 2, "coare3p6", 2.0, 10.0, *args, Niter=6, l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
    assert r1["init_report"]["hum_type"] == "sh" and r1["init_report"]["n_masked"] == 0
    r2 = ab.aerobulk_model(2, 2, "coare3p6", 2.0, 10.0, *args, l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])  # Niter sticky
    s = oracle.OracleSession("coare3p6", ni * nj, 2, True)
    sens = sensitivity(oracle, "coare3p6", True, 2.0, 10.0, 6, f, nt=2)
    for jt, r in ((1, r1), (2, r2)):
        o = s.compute(jt, 2.0, 10.0, 6, *[f[k] for k in IN6], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        got = {k: r[c].ravel(order="F") for k, c in (("ql", "QL"), ("qh", "QH"), ("tau_x", "Tau_x"), ("tau_y", "Tau_y"),
                                                      ("evap", "Evap"), ("t_s", "T_s"))}
        assert r["QL"].shape == (ni, nj)
        assert_hot_parity(got, o, ("ql", "qh", "tau_x", "tau_y", "evap", "t_s"), sens=sens, jt=jt, label=f"model jt={jt}")
    # radiation given but no skin: T_s is returned and equals sst
    r = ab.aerobulk_model(1, 1, "ecmwf", 2.0, 10.0, *args, Niter=5, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
    np.testing.assert_array_equal(r["T_s"], F["sst"])
    o = oracle.OracleSession("ecmwf", ni * nj).compute(1, 2.0, 10.0, 5, *[f[k] for k in IN6])
    assert_hot_parity({"ql": r["QL"].ravel(order="F")}, o, ("ql",), sens=sensitivity(oracle, "ecmwf", False, 2.0, 10.0, 5, f),
                      label="ecmwf no skin w/ rad")


def test_init_checks_on_gpu_match_reference_conditions(oracle):
    """AEROBULK_INIT: mask count, humidity-type detection and the STOP conditions (mod_aerobulk.f90:105-153)."""
    import aerobulk_amd as ab
    f = oracle.synth_fields(120, 50)
    n = f["sst"].size
    base = [f[k] for k in IN6]

    def init(fields, **kw):
        with ab.Session("ncar", n) as s:
            return s.init(*fields, **kw)

    ok = init(base)
    ref = oracle.init_checks(*base)
    assert ok["hum_type"] == ref["hum_type"] == "sh" and ok["n_masked"] == ref["n_masked"] == 0
    few = [a.copy() for a in base]
    few[0][:7] = 0.0                # silly SST on 7 cells -> masked, not fatal
    few[5][100:103] = 0.0           # silly SLP on 3 cells
    r = init(few)
    assert r["n_masked"] == oracle.init_checks(*few)["n_masked"] == 10
    rh = list(base); rh[2] = np.full(n, 80.0)
    assert init(rh)["hum_type"] == "rh"
    dpt = list(base); dpt[2] = f["t_zt"] - 2.0
    assert init(dpt)["hum_type"] == "dp"
    cel = list(base); cel[0] = f["sst"] - 273.15
    with pytest.raises(ab.AerobulkError) as e:
        init(cel)
    assert e.value.status == 5 and "whole domain is masked" in e.value.message
    hum = list(base); hum[2] = np.full(n, 500.0)
    with pytest.raises(ab.AerobulkError) as e:
        init(hum)
    assert e.value.status == 6
    # u10 beyond +-50 m/s on a cell whose wind module is masked... the mean/min/max test must still use masked stats
    u = [a.copy() for a in base]; u[3][5] = 60.0
    assert init(u)["n_masked"] == 1
    # radiation fields: the reference checks rad_lw twice (prsw=rad_lw), so a silly rad_sw is NOT detected
    with ab.Session("coare3p6", n, 1, 1, True) as s:
        rep = s.init(*base, rad_sw=f["rad_lw"], rad_lw=f["rad_lw"])
        assert rep["n_masked"] == 0
    lw = f["rad_lw"].copy(); lw[:4] = 5000.0
    with ab.Session("coare3p6", n, 1, 1, True) as s:
        assert s.init(*base, rad_sw=lw, rad_lw=lw)["n_masked"] == 4


@pytest.mark.parametrize("world,extra", [(2, []), (2, ["--peer-rows", "-1", "--no-early-gather"]), (3, ["--peer-rows", "40", "--gather-ts"]),
                                         (2, ["--no-pipeline-gather"]), (3, ["--steps", "5", "--peer-rows", "60"]),
                                         (4, []), (2, ["--config", "4"]), (3, ["--config", "4", "--skin", "--no-early-gather", "--peer-rows", "50"]),
                                         (2, ["--config", "5"]), (2, ["--config", "5", "--no-early-gather", "--peer-rows", "-1"]),
                                         (2, ["--config", "2", "--peer-rows", "100"]), (2, ["--config", "1"])])
def test_bench_sharded_path_several_ranks_one_gpu(world, extra):
    """bench.py's N>1 path (root-heavy j-block sharding measured or given, row chunks, packed gather joined by rank 0 before or
    after its own compute, reassembly; BASELINE configs 2-5) run as several ranks that share the one visible GPU, with the gloo
    backend standing in for RCCL (RCCL refuses two ranks per device) in the same post / compute / wait order.  --verify makes
    rank 0 recompute the whole grid alone and demand bit-identical gathered fields, for every algorithm of the step."""
    import sys
    import torch
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + world + 7 * len(extra) + len("".join(extra))), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "3", "--warmup", "1", "--grid", "720x333", "--backend", "gloo", "--verify", "--chunks", "3", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == world and res["verify"].startswith("gathered == single-GPU")
    assert res["scaling"] == "strong" and res["value"] > 0
    assert "rank 0 owns" in res["config"]["sharding"]
    assert ("after computing" if "--no-early-gather" in extra else "before computing") in res["config"]["sharding"]
    assert res["resident"]["value"] > 0                      # the second number: fluxes left distributed
    # two buffer sets in flight (consecutive steps of a --verify run compute different fields: a gather through the wrong set fails the
    # bit-identity above); both orders are reported
    assert res["gather"]["pipelined"] == ("--no-pipeline-gather" not in extra)
    if res["gather"]["pipelined"]:
        assert res["gather"]["unpipelined"]["value"] > 0
    assert res["roofline"]["bound"] == "hbm" and res["roofline"]["limiter"] == "valu_fp64"   # the HBM figures BASELINE.json asks for; what binds is VALU issue
    assert res["limiter"].startswith("valu_fp64") and res["calib"]["fma_f64_tflops_after"] > 30 and res["value_norm"] > 0
    if not extra:
        assert res["host_path"]["ms_per_record"] > 0 and str(world) + " ranks" in res["host_path"]["layout"]      # every rank stages its own rows
        assert res["overlapped"]["value"] > 0 and "every rank its whole block" in res["overlapped"]["note"]
        assert res["gather"]["model"]["inputs"]["n_gpus"] == world and len(res["gather"]["model"]["predicted"]) == 4
    if "--config" in extra and extra[extra.index("--config") + 1] == "4":
        assert len(res["per_algorithm"]) == 5 and all(v["Mcell_per_s"] > 0 for v in res["per_algorithm"].values())
    if "--config" in extra and extra[extra.index("--config") + 1] == "5":
        assert "AB_F32_MIXED" in res["dtype"] and "ecmwf + cool-skin" in res["config"]["workload"]   # config 5 is timed on the mixed mode
    if not extra:
        assert "link_GBps" in res["config"]["sharding_tuning"], res["config"]


@pytest.mark.parametrize("nsh,extra,env", [(2, [], {}), (2, ["--gather-ts"], {"AEROBULK_AMD_GATHER": "rccl"}), (2, ["--config", "4", "--no-pipeline-gather"], {}),
                                           (8, [], {"AEROBULK_AMD_GATHER": "rccl"}), (4, ["--peer-rows", "40"], {}), (3, ["--peer-rows", "-1"], {})],
                         ids=["d2d", "rccl-one-device-communicator", "config4-unpipelined", "eight-shards-rccl-measured-split", "four-shards-given-split",
                              "three-equal-shards"])
def test_bench_as_typed_runs_the_library_sharded_session(nsh, extra, env):
    """`python3 bench.py --gpus N` exactly as typed (no launcher, no WORLD_SIZE): one process, one sharded library session —
    ab_session_compute_shards on device-resident rows + ab_session_gather — here with every shard on device 0 (--devices 0,0,...: the
    box has one GPU); AEROBULK_AMD_GATHER=rccl makes the shards' rows travel by ncclSend / ncclRecv on a one-device communicator.
    The cut is root-heavy (ab_session_create_sharded_rows): measured during set-up by default, --peer-rows gives it, -1 = equal blocks.
    --verify: the gathered fields are bit-identical to one unsharded session."""
    import sys
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nsh), "--devices", ",".join(["0"] * nsh), "--grid", "720x333", "--steps", "3",
           "--warmup", "1", "--verify", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == nsh and res["launcher"] == "inprocess" and res["verify"].startswith("gathered == single-GPU")
    assert res["value"] > 0 and res["resident"]["value"] > 0 and len(res["per_device_kernel_ms"]) == nsh
    assert res["n_ranks_rccl"] == (1 if env else 0)
    assert res["roofline"]["achieved"] > 0
    # round 6: the line carries what lets the first run on N real devices be judged — the box's calibration, the one-GPU step of the same run,
    # the gathered model with its inputs and predictions, and (one pass per step) the host-array leg through the sharded session
    assert res["calib"]["fma_f64_tflops_before"] > 30 and res["value_norm"] > 0 and res["limiter"].startswith("valu_fp64")
    assert res["one_gpu"]["ms_per_step"] > 0 and res["speedup_vs_one_gpu"]["resident"] > 0
    gm = res["gather"]["model"]
    assert gm["inputs"]["n_gpus"] == nsh and gm["inputs"]["bytes_per_cell_on_link"] in (40, 48, 200) and len(gm["predicted"]) >= 3
    assert all(0 < v["peer_share"] <= 1.0 / nsh + 1e-9 and v["speedup_vs_one_gpu"] > 0 for v in gm["predicted"].values()) and gm["gather_free_ceiling"] > 0
    if "--config" not in extra:
        hp = res["host_path"]
        assert hp["ms_per_record"] > 0 and hp["first_record_ms"] >= hp["ms_per_record"] and hp["Mcell_s"] > 0 and hp["bytes_per_cell_over_pcie"] == 112
    rows = res["rows_per_shard"]
    assert len(rows) == nsh and sum(rows) == 333 and min(rows) >= 1
    if "--peer-rows" in extra and extra[extra.index("--peer-rows") + 1] == "-1":
        assert res["split"] == "equal j-blocks" and max(rows) - min(rows) <= 1
    elif "--peer-rows" in extra:
        assert rows[1:] == [40] * (nsh - 1) and rows[0] == 333 - 40 * (nsh - 1) and "root-heavy" in res["split"]
    else:
        assert "root-heavy" in res["split"] and len(set(rows[1:])) == 1 and rows[0] >= rows[1] and "kernel_Mcell_per_s_per_device" in res["split_tuning"]


def test_sharded_init_statistics_match_global_init(oracle):
    """AEROBULK_INIT on a grid sharded over ranks (SURVEY §8e "the one true exchange"): per-shard ab_session_init_stats,
    SUM/MIN/MAX combination (what an all-reduce does), ab_session_init_apply -> same report as one global init."""
    import aerobulk_amd as ab
    ni, nj = 200, 90
    f = oracle.synth_fields(ni, nj)
    fields = [f[k].copy() for k in IN6]
    fields[0][[3, 4000, 17000]] = 0.0      # silly SST in both shards
    fields[5][9000:9005] = 0.0             # silly SLP in the second shard
    n = ni * nj
    cut = ni * 41                          # ragged j-blocks: 41 + 49 rows
    with ab.Session("coare3p6", n, 1, 1, True) as g:
        want = g.init(*fields, rad_sw=f["rad_lw"], rad_lw=f["rad_lw"])
    parts = []
    sessions = [ab.Session("coare3p6", cut, 1, 1, True), ab.Session("coare3p6", n - cut, 1, 1, True)]
    for s, sl in zip(sessions, (slice(0, cut), slice(cut, n))):
        parts.append(s.init_stats(*[a[sl] for a in fields], rad_sw=f["rad_lw"][sl], rad_lw=f["rad_lw"][sl]))
    st = np.concatenate([np.sum([p[0:11] for p in parts], axis=0), np.min([p[11:20] for p in parts], axis=0),
                         np.max([p[20:29] for p in parts], axis=0)])
    for s in sessions:
        got = s.init_apply(st, have_rad=True)
        assert (got["n_cells"], got["n_masked"], got["hum_type"]) == (want["n_cells"], want["n_masked"], want["hum_type"]) == (n, 8, "sh")
        s.close()


def test_model_first_record_with_fused_init_equals_two_pass(oracle):
    """AEROBULK_MODEL at jt == 1 on a grid large enough for the chunk pipeline: AEROBULK_INIT's statistics ride on the pipelined
    pass (kernels start on the FIRST chunk's verdict about the humidity type).  Same results and same report as statistics in a
    pass of their own; a domain whose first chunk looks like specific humidity but which is relative humidity as a whole is
    recomputed with the global verdict; an all-masked domain is still an error."""
    import aerobulk_amd as ab
    ni, nj = 2200, 2001                                   # 4.4 M cells: 5 chunks
    f = oracle.synth_fields(ni, nj)
    F = {k: v.reshape((ni, nj), order="F") for k, v in f.items()}
    rh = 30.0 + 60.0 * f["rad_sw"] / 900.0               # 30 .. 90 %
    rh[: 1 << 20] = 0.05                                  # the first chunk alone reads as 'sh' (all < 0.08): the domain is 'rh'
    cases = {"sh": F["hum_zt"], "rh-behind-sh": rh.reshape((ni, nj), order="F")}
    for name, hum in cases.items():
        res = {}
        for mode in ("fused", "two-pass"):
            os.environ.pop("AEROBULK_AMD_NO_FUSED_INIT", None)
            if mode == "fused":
                os.environ["AEROBULK_AMD_FUSED_INIT"] = "1"      # opt-in since round 4: the default is the reference's order
            else:
                os.environ.pop("AEROBULK_AMD_FUSED_INIT", None)
            try:
                r = ab.aerobulk_model(1, 1, "coare3p6", 2.0, 10.0, F["sst"], F["t_zt"], hum, F["u_zu"], F["v_zu"], F["slp"], Niter=4,
                                      l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
            finally:
                os.environ.pop("AEROBULK_AMD_FUSED_INIT", None)
            res[mode] = r
        a, b = res["fused"], res["two-pass"]
        assert a["init_report"] == b["init_report"] and a["init_report"]["hum_type"] == ("sh" if name == "sh" else "rh"), a["init_report"]
        for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{name} {k}")
    with pytest.raises(ab.AerobulkError) as e:            # Celsius SST: the whole domain is masked
        ab.aerobulk_model(1, 1, "coare3p6", 2.0, 10.0, F["sst"] - 273.15, F["t_zt"], F["hum_zt"], F["u_zu"], F["v_zu"], F["slp"], Niter=4)
    assert e.value.status == 5


def test_init_errors_at_jt1_leave_the_callers_arrays_untouched(oracle):
    """The reference's order at jt == 1 (mod_aerobulk.f90:246-262): AEROBULK_INIT aborts BEFORE anything is computed.  On a grid large
    enough for the chunk pipeline (>= 4 Mi cells), through the C ABI's ab_model with caller-owned output arrays: after
    AB_ERR_HUM_TYPE and AB_ERR_ALL_MASKED every output array still holds the caller's sentinel; the opt-in one-pass first record
    (AEROBULK_AMD_FUSED_INIT=1) returns the same codes.  (AB_ERR_UNITS cannot be produced by fields: the mask of
    mod_aerobulk.f90:108-115 uses the very ranges check_unit_consistency tests, and the humidity's range is that of the detected
    type; ab_session_init_apply's branch is exercised with doctored statistics in test_abi / test_gpu_sharded.)"""
    import ctypes as C
    from aerobulk_amd import _lib
    L = _lib.load()
    ni, nj = 2200, 2001                                   # 4.4 M cells
    f = oracle.synth_fields(ni, nj)
    n = ni * nj
    dp = C.POINTER(C.c_double)
    sentinel = -777.25
    outs = [np.full(n, sentinel) for _ in range(6)]
    alg = b"coare3p6"

    def model(fields):
        p = {k: np.ascontiguousarray(v).ctypes.data_as(dp) for k, v in fields.items()}
        o = [a.ctypes.data_as(dp) for a in outs]
        return L.ab_model(1, 1, alg, len(alg), 2.0, 10.0, p["sst"], p["t_zt"], p["hum_zt"], p["u_zu"], p["v_zu"], p["slp"], o[0], o[1], o[2],
                          o[3], o[4], 4, 1, p["rad_sw"], p["rad_lw"], o[5], ni, nj, None)

    cases = ((6, dict(f, hum_zt=np.full(n, 500.0))),       # AB_ERR_HUM_TYPE: neither kg/kg, K nor %
             (5, dict(f, sst=f["sst"] - 273.15)))          # AB_ERR_ALL_MASKED: SST in Celsius
    for want, fields in cases:
        assert model(fields) == want, L.ab_last_error()
        for i, o in enumerate(outs):
            assert np.all(o == sentinel), (want, i)        # nothing was written
    os.environ["AEROBULK_AMD_FUSED_INIT"] = "1"
    try:
        for want, fields in cases:
            assert model(fields) == want, L.ab_last_error()
    finally:
        os.environ.pop("AEROBULK_AMD_FUSED_INIT", None)
    assert model(f) == 0                                   # and the good fields compute
    assert not np.any(outs[0] == sentinel)
