"""GPU: row-block sharding INSIDE the library (ab_session_create_sharded / AB_DEVICE_ALL / AEROBULK_AMD_DEVICES), exercised on
one GPU with k shards on it: every result must be bit-identical to the unsharded session (pointwise path), the
AEROBULK_INIT statistics — the one exchange of the path — must combine to the global report, errors must come through."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
IN6 = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")
OUT = ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")


def _run(ab, f, device, algo="coare3p6", skin=True, nt=3, ni=360, nj=181, niter=5, to_device=False):
    import torch
    conv = (lambda a: torch.from_numpy(a).cuda()) if to_device else (lambda a: a)
    ins = [conv(f[k]) for k in IN6]
    rs, rl = (conv(f["rad_sw"]), conv(f["rad_lw"])) if skin else (None, None)
    recs = []
    with ab.Session(algo, ni, nj, nt, skin, device=device) as s:
        rep = s.init(*ins, rad_sw=rl, rad_lw=rl)
        for jt in range(1, nt + 1):
            o = s.compute(jt, 2.0, 10.0, *ins, Niter=niter, rad_sw=rs, rad_lw=rl)
            recs.append({k: (v.cpu().numpy() if to_device else np.array(v)) for k, v in o.items()})
        wl = s.wl_state() if (skin and nt > 1) else None
        shards = s.shards()
    return rep, recs, wl, shards


@pytest.mark.parametrize("to_device", [False, True], ids=["host-arrays", "device-arrays"])
@pytest.mark.parametrize("algo,skin", [("coare3p6", True), ("ecmwf", True), ("ncar", False)])
def test_k_shards_on_one_gpu_are_bit_identical(oracle, algo, skin, to_device):
    import aerobulk_amd as ab
    ni, nj = 360, 181                                   # 181 rows over 3 shards: 61 + 60 + 60
    f = oracle.synth_fields(ni, nj)
    nt = 3 if skin else 1
    rep1, one, wl1, sh1 = _run(ab, f, 0, algo, skin, nt, ni, nj, to_device=to_device)
    repk, many, wlk, shk = _run(ab, f, [0, 0, 0], algo, skin, nt, ni, nj, to_device=to_device)
    assert sh1 == [(0, nj, 0)] and shk == [(0, 61, 0), (61, 60, 0), (121, 60, 0)]
    assert rep1 == repk and rep1["hum_type"] == "sh" and rep1["n_masked"] == 0
    for jt, (a, b) in enumerate(zip(one, many), 1):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{algo} jt={jt} {k}")
    if wl1 is not None:
        for k in wl1:
            np.testing.assert_array_equal(wl1[k], wlk[k], err_msg=k)


def test_sharded_pipelined_host_path_large_grid(oracle):
    """Each shard >= 4 Mi cells: every shard takes the chunk-pipelined host path (its own three streams + drain thread),
    concurrently with the others; AB_DEVICE_ALL on a one-GPU box is an ordinary session."""
    import aerobulk_amd as ab
    ni, nj = 2200, 4003                                  # 8.8 M cells, two shards of 2002 / 2001 rows
    f = oracle.synth_fields(ni, nj)
    ins = [f[k] for k in IN6]
    with ab.Session("coare3p6", ni, nj, 1, True, device="all") as s:
        assert len(s.shards()) == max(1, ab.device_count())
        one = s.compute(1, 2.0, 10.0, *ins, Niter=4, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
    with ab.Session("coare3p6", ni, nj, 1, True, device=[0, 0]) as s:
        two = s.compute(1, 2.0, 10.0, *ins, Niter=4, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
        assert s.last_kernel_ms() > 0
    for k in one:
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)


def test_sharded_init_conditions_and_errors(oracle):
    """AEROBULK_INIT decisions on the combined statistics: a masked cell in ONE shard is counted once, a humidity field that only
    looks like 'rh' in one shard is judged on the whole domain, a 10 N/m2 cell in one shard fails the record."""
    import aerobulk_amd as ab
    ni, nj = 128, 66
    f = oracle.synth_fields(ni, nj)
    n = ni * nj
    ins = {k: f[k].copy() for k in IN6}
    ins["sst"][n - 5] = 400.0                            # silly cell in the last shard: masked, not an error
    with ab.Session("ncar", ni, nj, 1, False, device=[0, 0, 0]) as s:
        rep = s.init(*[ins[k] for k in IN6])
        assert rep["n_masked"] == 1 and rep["hum_type"] == "sh"
    bad = {k: f[k].copy() for k in IN6}
    bad["hum_zt"][: n // 2] = 500.0                      # neither sh, dp nor rh once the halves are combined
    with ab.Session("ncar", ni, nj, 1, False, device=[0, 0]) as s:
        with pytest.raises(ab.AerobulkError) as e:
            s.init(*[bad[k] for k in IN6])
        assert e.value.status == 6
    wild = {k: f[k].copy() for k in IN6}
    wild["u_zu"][3] = 49.0; wild["v_zu"][3] = 0.0; wild["t_zt"][3] = wild["sst"][3] - 12.0   # first shard only
    with ab.Session("coare3p6", ni, nj, 1, False, device=[0, 0, 0]) as s:
        try:
            s.compute(1, 2.0, 10.0, *[wild[k] for k in IN6], Niter=5)
            tripped = False
        except ab.AerobulkError as e:
            tripped = e.status == 8
    with ab.Session("coare3p6", ni, nj, 1, False) as s:
        try:
            s.compute(1, 2.0, 10.0, *[wild[k] for k in IN6], Niter=5)
            tripped1 = False
        except ab.AerobulkError as e:
            tripped1 = e.status == 8
    assert tripped == tripped1


def test_device_arrays_on_another_gpu_are_refused():
    import torch
    import aerobulk_amd as ab
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ni, nj = 64, 8
    f = ab.synth_fields_device(ni, nj, device=torch.device("cuda", 0))
    with ab.Session("ncar", ni, nj, 1, False, device=[0, 1]) as s:
        with pytest.raises(ab.AerobulkError) as e:
            s.compute(1, 2.0, 10.0, *[f[k] for k in ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")])
        assert e.value.status == 10


def test_aerobulk_model_over_shards_through_the_environment(oracle, tmp_path):
    """AEROBULK_MODEL (process-global session, the reference's calling convention) spread over shards by AEROBULK_AMD_DEVICES,
    no change to the caller: same results as unset, warm-layer state carried per shard over jt = 1..3."""
    script = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
from oracle import pyoracle as po
ni, nj = 96, 50
f = po.synth_fields(ni, nj)
F = {k: v.reshape((ni, nj), order="F") for k, v in f.items()}
args = [F[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")]
out = []
for jt in (1, 2, 3):
    r = ab.aerobulk_model(jt, 3, "coare3p6", 2.0, 10.0, *args, Niter=5, l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
    out.append(np.stack([r[k] for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")]))
np.save(sys.argv[2], np.stack(out))
'''
    res = {}
    for tag, env in (("one", None), ("three", "0,0,0"), ("count", "1")):
        e = dict(os.environ)
        e.pop("AEROBULK_AMD_DEVICES", None)
        if env:
            e["AEROBULK_AMD_DEVICES"] = env
        out = str(tmp_path / f"{tag}.npy")
        pr = subprocess.run([sys.executable, "-c", script, ROOT, out], env=e, capture_output=True, text=True, timeout=600)
        assert pr.returncode == 0, pr.stdout + pr.stderr
        res[tag] = np.load(out)
    np.testing.assert_array_equal(res["one"], res["three"])
    np.testing.assert_array_equal(res["one"], res["count"])
    assert np.abs(res["one"][0, 5] - res["one"][2, 5]).max() > 1e-6      # the warm layer did evolve


def test_sharded_turb_and_diagnostics(oracle):
    """TURB_COARE3P6 with both skin schemes and the OPTIONAL outputs through a sharded session: slices of the caller's arrays."""
    import aerobulk_amd as ab
    ni, nj = 200, 31
    n = ni * nj
    f = oracle.synth_fields(ni, nj)
    theta = f["t_zt"] + 0.02
    ssq = 0.98 * np.array([oracle.lib().abo_q_sat(t, p) for t, p in zip(f["sst"], f["slp"])])
    qsw = 0.934 * f["rad_sw"]

    def run(device):
        with ab.Session("coare3p6", ni, nj, 1, False, device=device) as s:
            d = s.set_diagnostics(["u_star", "dT_cs", "Hz_wl"])
            T_s, q_s = f["sst"].copy(), ssq.copy()
            o = s.turb(1, 2.0, 10.0, T_s, theta, q_s, f["hum_zt"], np.hypot(f["u_zu"], f["v_zu"]), True, True, qsw, f["rad_lw"], f["slp"], nb_iter=5)
            return {**{k: np.array(v) for k, v in o.items()}, **{k: np.array(v) for k, v in d.items()}, "T_s": T_s, "q_s": q_s}

    a, b = run(0), run([0, 0, 0, 0])
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.abs(a["T_s"] - f["sst"]).max() > 0.05


def _rank_worker(rank, world, port, ni, nj, out_path):
    """One rank of a 2-process run (gloo rendezvous, both ranks on the one GPU of the test box): its half of the rows goes through
    a SHARDED session of two shards; AEROBULK_INIT's statistics are all-reduced between the ranks (SUM / MIN / MAX)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import aerobulk_amd as ab
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = -(-nj // world)
    j0 = rank * per
    njl = min(per, nj - j0)
    f = po.synth_fields(ni, nj, j0, njl)
    ins = [f[k] for k in IN6]
    with ab.Session("coare3p6", ni, njl, 2, True, device=[0, 0]) as s:
        st = torch.from_numpy(s.init_stats(*ins, rad_sw=f["rad_lw"], rad_lw=f["rad_lw"]))
        a, b, c = st[0:11].clone(), st[11:20].clone(), st[20:29].clone()
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        dist.all_reduce(b, op=dist.ReduceOp.MIN)
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        rep = s.init_apply(torch.cat([a, b, c]).numpy(), have_rad=True)
        assert rep["n_cells"] == ni * nj and rep["hum_type"] == "sh", rep
        recs = []
        for jt in (1, 2):
            o = s.compute(jt, 2.0, 10.0, *ins, Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
            recs.append(np.stack([o[k] for k in OUT]))
    mine = torch.from_numpy(np.stack(recs))                       # [2, 6, n_local]
    pad = torch.zeros((2, 6, ni * per), dtype=torch.float64)
    pad[:, :, :ni * njl] = mine
    gl = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, gl, dst=0)
    if rank == 0:
        parts = [gl[r][:, :, :ni * min(per, nj - r * per)].numpy() for r in range(world)]
        np.save(out_path, np.concatenate(parts, axis=2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_each_with_a_sharded_session(oracle, tmp_path):
    import torch.multiprocessing as mp
    import aerobulk_amd as ab
    ni, nj, world = 160, 45, 2
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "glob.npy")
    port = 23000 + (os.getpid() % 4000)
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, ni, nj, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    glob = np.load(out)
    f = oracle.synth_fields(ni, nj)
    with ab.Session("coare3p6", ni, nj, 2, True) as s:
        for jt in (1, 2):
            o = s.compute(jt, 2.0, 10.0, *[f[k] for k in IN6], Niter=5, rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
            for i, k in enumerate(OUT):
                np.testing.assert_array_equal(glob[jt - 1, i], o[k], err_msg=f"jt={jt} {k}")


# ---------------------------------------------------------------------------------------------------------------------------------
# Device-resident shards and the gather of the fluxes (include/aerobulk_amd.h: ab_session_compute_shards / ab_session_gather): the
# north_star's layout — rows sharded across GPUs, fields resident, "a trivial RCCL gather of the output tau/Q_L/Q_H/E arrays".  The
# test box has ONE GPU: k shards on it exercise the shard bookkeeping, the per-shard launches, the placement of every shard's rows and
# the stream ordering; AEROBULK_AMD_GATHER=rccl sends the same rows through ncclSend / ncclRecv on a one-device communicator (a rank
# talking to itself).  No multi-device run has happened.
def _shards_and_gather(ab, oracle, algo, skin, nsh, root, gather_ts, precision="f64", rows=None):
    import torch
    ni, nj, nt = 192, 100, 2
    f = oracle.synth_fields(ni, nj)
    dt = torch.float64 if precision == "f64" else torch.float32
    names = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
    keys = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp", "rad_sw", "rad_lw")
    whole = {k: torch.from_numpy(f[n]).to(dt).cuda() for n, k in zip(names, keys)}
    outs = ("QL", "QH", "Tau_x", "Tau_y", "Evap") + (("T_s",) if skin else ())
    with ab.Session(algo, ni, nj, nt, skin, precision=precision) as s:
        ref = []
        for jt in range(1, nt + 1):
            o = s.compute(jt, 2.0, 10.0, *[whole[k] for k in keys[:6]], Niter=5, rad_sw=whole["rad_sw"] if skin else None,
                          rad_lw=whole["rad_lw"] if skin else None)
            ref.append({k: v.clone() for k, v in o.items()})
    dev_before = torch.cuda.current_device()
    with ab.Session(algo, ni, nj, nt, skin, precision=precision, device=[0] * nsh, rows=rows) as s:
        assert torch.cuda.current_device() == dev_before          # round-2 advisory: the caller's device is the caller's
        sh = s.shards()
        if rows is not None:
            assert [x[1] for x in sh] == list(rows) and [x[0] for x in sh] == [sum(rows[:i]) for i in range(nsh)]
        fields = [{k: whole[k][j0 * ni:(j0 + njl) * ni].clone() for k in (keys if skin else keys[:6])} for j0, njl, _ in sh]
        for jt in range(1, nt + 1):
            shard_out = [{k: torch.empty(ni * njl, dtype=dt, device="cuda") for k in outs} for _, njl, _ in sh]
            s.compute_shards(jt, 2.0, 10.0, fields, shard_out, Niter=5)
            want = outs if gather_ts else outs[:5]
            dst = {k: torch.full((ni * nj,), float("nan"), dtype=dt, device="cuda") for k in want}
            s.gather(shard_out, dst, root=root)
            for k in want:
                assert torch.equal(dst[k], ref[jt - 1][k]), (algo, jt, k)
        assert torch.cuda.current_device() == dev_before


@pytest.mark.parametrize("algo,skin,nsh,root,gather_ts", [("coare3p6", True, 3, 0, True), ("ecmwf", True, 4, 2, False), ("ncar", False, 2, 1, False),
                                                          ("coare3p6", True, 1, 0, True)])
def test_device_resident_shards_and_gather(oracle, algo, skin, nsh, root, gather_ts):
    import aerobulk_amd as ab
    _shards_and_gather(ab, oracle, algo, skin, nsh, root, gather_ts)


def test_unequal_row_blocks_and_their_gather(oracle):
    """ab_session_create_sharded_rows: the caller's row counts (the gather's destination owns more rows than its peers); compute,
    warm-layer state over two records and the gather land every shard's rows in their place; bad counts are refused."""
    import aerobulk_amd as ab
    _shards_and_gather(ab, oracle, "coare3p6", True, 4, 0, True, rows=[58, 14, 14, 14])
    _shards_and_gather(ab, oracle, "ecmwf", True, 3, 2, False, rows=[1, 98, 1])
    for rows in ([50, 49], [0, 100], [60, 60]):
        with pytest.raises(ab.AerobulkError) as e:
            ab.Session("ncar", 192, 100, 1, False, device=[0, 0], rows=rows)
        assert e.value.status == 10


def test_gather_through_rccl_on_a_one_device_communicator(oracle, tmp_path):
    """The RCCL leg of ab_session_gather — dlopen of librccl, ncclCommInitAll, one group of ncclSend / ncclRecv per gather straight
    into the rows of the destination — on the one GPU of the box: AEROBULK_AMD_GATHER=rccl makes every shard but the root's own travel
    that way (rank 0 to rank 0).  fp64 and fp32 arrays; same bits as the unsharded session."""
    script = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import aerobulk_amd as ab
from oracle import pyoracle as po
from test_gpu_sharded import _shards_and_gather
_shards_and_gather(ab, po, "coare3p6", True, 3, 1, True)
_shards_and_gather(ab, po, "ecmwf", True, 2, 0, False, precision="f32_mixed")
_shards_and_gather(ab, po, "coare3p6", True, 8, 0, False, rows=[30] + [10] * 7)      # eight shards, root-heavy cut: 7 x 5 sends / receives in one group
print("GATHER_OK")
'''
    e = dict(os.environ, AEROBULK_AMD_GATHER="rccl")
    pr = subprocess.run([sys.executable, "-c", script, ROOT], env=e, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0 and "GATHER_OK" in pr.stdout, pr.stdout[-2000:] + pr.stderr[-4000:]


def test_first_record_of_aerobulk_model_is_fused_for_shards_too(oracle, tmp_path):
    """AEROBULK_MODEL at jt == 1 through AEROBULK_AMD_DEVICES=0,0 on a grid large enough for the pipelined path (each shard >= 4 Mi
    cells): AEROBULK_INIT's statistics ride on every shard's pass, the verdict is taken on the COMBINED statistics, and a shard whose
    first chunk misjudged the humidity type computes its rows again from its resident fields — here the domain is relative humidity
    and the first 2^20 cells of shard 1 alone read as specific humidity.  Same results and the same report as the two-pass path
    (the default since round 4: the reference's order) and as one device.  The one-pass first record is the opt-in
    AEROBULK_AMD_FUSED_INIT=1."""
    script = r'''
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
from oracle import pyoracle as po
ni, nj = 2048, 4200                       # 8.6 M cells: two shards of 2100 rows = 4.3 M cells each
f = po.synth_fields(ni, nj)
rh = 30.0 + 60.0 * f["rad_sw"] / 900.0    # 30 .. 90 %
o1 = ni * 2100
rh[o1: o1 + (1 << 20)] = 0.05             # the first chunk of shard 1 alone reads as 'sh' (all < 0.08)
F = {k: v.reshape((ni, nj), order="F") for k, v in dict(f, hum_zt=rh).items()}
args = [F[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")]
r = ab.aerobulk_model(1, 1, "coare3p6", 2.0, 10.0, *args, Niter=4, l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
np.save(sys.argv[2], np.stack([r[k] for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")]))
print("REPORT " + json.dumps(r["init_report"]))
'''
    res, reps = {}, {}
    for tag, extra in (("fused", {"AEROBULK_AMD_FUSED_INIT": "1"}), ("twopass", {}), ("one", None)):
        e = dict(os.environ)
        e.pop("AEROBULK_AMD_DEVICES", None)
        e.pop("AEROBULK_AMD_NO_FUSED_INIT", None)
        e.pop("AEROBULK_AMD_FUSED_INIT", None)
        if extra is not None:
            e["AEROBULK_AMD_DEVICES"] = "0,0"
            e.update(extra)
        out = str(tmp_path / f"{tag}.npy")
        pr = subprocess.run([sys.executable, "-c", script, ROOT, out], env=e, capture_output=True, text=True, timeout=900)
        assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
        res[tag] = np.load(out)
        reps[tag] = [ln for ln in pr.stdout.splitlines() if ln.startswith("REPORT ")][0]
        if tag == "fused":
            assert "computed again" in pr.stderr          # the recompute is announced (round-2 advisory)
    assert reps["fused"] == reps["twopass"] == reps["one"] and '"hum_type": "rh"' in reps["fused"], reps
    np.testing.assert_array_equal(res["fused"], res["twopass"])
    np.testing.assert_array_equal(res["fused"], res["one"])


def test_fused_first_record_needs_every_shard_to_qualify(oracle, tmp_path):
    """Round-3 advisory: with nj = 4095 over two shards the first shard has 2048 rows (4 194 304 cells = the pipelining threshold)
    and the second 2047 (just under it).  The opt-in fused first record must not be taken (a shard that cannot pipeline takes no
    statistics): the session falls back to the reference's order, same results and report as one device.  Under the default order
    the caller's arrays are untouched when AEROBULK_INIT rejects the fields."""
    script = r'''
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
from oracle import pyoracle as po
ni, nj = 2048, 4095
f = po.synth_fields(ni, nj)
F = {k: v.reshape((ni, nj), order="F") for k, v in f.items()}
args = [F[k] for k in ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp")]
r = ab.aerobulk_model(1, 1, "coare3p6", 2.0, 10.0, *args, Niter=4, l_use_skin=True, rad_sw=F["rad_sw"], rad_lw=F["rad_lw"])
np.save(sys.argv[2], np.stack([r[k] for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")]))
print("REPORT " + json.dumps(r["init_report"]))
'''
    res, reps = {}, {}
    for tag, extra in (("fused-asked", {"AEROBULK_AMD_DEVICES": "0,0", "AEROBULK_AMD_FUSED_INIT": "1"}), ("one", {})):
        e = dict(os.environ)
        for k in ("AEROBULK_AMD_DEVICES", "AEROBULK_AMD_NO_FUSED_INIT", "AEROBULK_AMD_FUSED_INIT"):
            e.pop(k, None)
        e.update(extra)
        out = str(tmp_path / f"{tag}.npy")
        pr = subprocess.run([sys.executable, "-c", script, ROOT, out], env=e, capture_output=True, text=True, timeout=900)
        assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
        res[tag] = np.load(out)
        reps[tag] = [ln for ln in pr.stdout.splitlines() if ln.startswith("REPORT ")][0]
    assert reps["fused-asked"] == reps["one"] and '"hum_type": "sh"' in reps["one"], reps
    np.testing.assert_array_equal(res["fused-asked"], res["one"])
