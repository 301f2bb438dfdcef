"""CPU, world_size 2 / 4 / 8 over gloo: the j-block sharding + packed-gather assembly that bench.py uses for N>1.
The per-rank compute is stood in by the CPU oracle (the HIP kernel needs a GPU; the path is pointwise, so the
sharding logic is independent of who computes a cell).  Checks that the gathered global field is bit-identical
to the single-process result — the property the multi-GPU run relies on (no halo)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ni, nj, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from bench import shard_rows
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    j0, njl, per = shard_rows(nj, world, rank)
    n_local, n_pad = ni * njl, ni * per
    f = po.synth_fields(ni, nj, j0, njl)
    o = po.OracleSession("coare3p6", n_local, 1, True).compute(
        1, 2.0, 10.0, 5, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
    names = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")
    buf = torch.zeros((6, n_pad), dtype=torch.float64)
    for i, k in enumerate(names):
        buf[i, :n_local] = torch.from_numpy(o[k])
    gl = [torch.empty((6, n_pad), dtype=torch.float64) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gl, dst=0)
    if rank == 0:
        glob = {}
        for i, k in enumerate(names):
            parts = []
            for r in range(world):
                _, njr, _ = shard_rows(nj, world, r)
                parts.append(gl[r][i, :ni * njr].numpy())
            glob[k] = np.concatenate(parts)
        q.put(glob)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nj", [(2, 40), (2, 41), (4, 9), (8, 37)])  # divisible, ragged, and ranks with few rows
def test_jblock_sharding_and_gather_reproduce_single_process(oracle, world, nj):
    import torch.multiprocessing as mp
    ni = 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + nj + 100 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, ni, nj, q)) for r in range(world)]
    for p in procs:
        p.start()
    glob = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = oracle.synth_fields(ni, nj)
    ref = oracle.OracleSession("coare3p6", ni * nj, 1, True).compute(
        1, 2.0, 10.0, 5, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"], rad_sw=f["rad_sw"], rad_lw=f["rad_lw"])
    for k in ("ql", "qh", "tau_x", "tau_y", "evap", "t_s"):
        np.testing.assert_array_equal(glob[k], ref[k], err_msg=k)


def test_shard_rows_partition():
    from bench import shard_rows
    for nj in (1, 7, 450, 3600, 10800):
        for world in (1, 2, 3, 4, 8):
            rows = [shard_rows(nj, world, r) for r in range(world)]
            assert sum(r[1] for r in rows) == nj
            pos = 0
            for j0, njl, per in rows:
                assert njl <= per
                if njl:
                    assert j0 == pos
                pos += njl


def test_root_heavy_sharding_and_balance_formula():
    """bench.py's gathered run: rank 0 (destination of the gather) owns more rows; the split equalises its compute with the peers'
    compute + transfer and falls back to the equal split when links are fast."""
    from bench import balanced_peer_rows, shard_rows_root_heavy
    for nj, world in ((3600, 8), (3600, 2), (37, 8), (41, 2), (8, 8)):
        for rp in (1, max(nj // world, 1), 10 ** 6):
            rows = [shard_rows_root_heavy(nj, world, r, rp) for r in range(world)]
            assert sum(r[1] for r in rows) == nj and rows[0][0] == 0
            pos = 0
            for j0, njl, per in rows:
                assert j0 == pos and njl >= 1
                pos += njl
            assert all(r[1] == rows[1][1] for r in rows[1:]) and rows[0][1] >= rows[1][1]
    t_cell = 3.5e-3 / 15.552e6
    assert balanced_peer_rows(3600, 8, t_cell, 40, 1e18) == 450                 # infinitely fast links: equal split
    slow = balanced_peer_rows(3600, 8, t_cell, 40, 60e9)
    assert 300 < slow < 450
    nj0 = 3600 - 7 * slow                                                          # both sides take about the same time
    assert abs(nj0 * t_cell - slow * 40 / 60e9) / (nj0 * t_cell) < 0.02


def _worker_pipelined(rank, world, port, ni, nj, nsteps, q):
    """bench.py's gathered time loop with TWO buffer sets: step t fills set t % 2 and posts its gathers without waiting; a set is
    written again only after its gathers of two steps ago were waited for (the order bench.py runs for RCCL and for gloo alike).
    Every step computes different fields (its own pass count), so a gather through the wrong set cannot go unnoticed."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from bench import shard_rows
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    j0, njl, per = shard_rows(nj, world, rank)
    n_local, n_pad = ni * njl, ni * per
    f = po.synth_fields(ni, nj, j0, njl)
    names = ("ql", "qh", "tau_x", "tau_y", "evap")
    sets = [torch.zeros((5, n_pad), dtype=torch.float64) for _ in range(2)]
    recv = [[torch.empty((5, n_pad), dtype=torch.float64) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
    inflight = [None, None]
    seen = {}

    def harvest(b, t):
        if inflight[b] is not None:
            inflight[b].wait()
            inflight[b] = None
            if rank == 0:
                seen[t] = [r.clone() for r in recv[b]]

    for t in range(nsteps):
        b = t % 2
        harvest(b, t - 2)
        o = po.OracleSession("coare3p6", n_local, 1, False).compute(1, 2.0, 10.0, 2 + t, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"])
        for i, k in enumerate(names):
            sets[b][i, :n_local] = torch.from_numpy(o[k])
        inflight[b] = dist.gather(sets[b], recv[b], dst=0, async_op=True)
    for t in (nsteps - 2, nsteps - 1):
        harvest(t % 2, t)
    if rank == 0:
        out = {}
        for t, parts in seen.items():
            out[t] = np.concatenate([parts[r][0, :ni * shard_rows(nj, world, r)[1]].numpy() for r in range(world)])
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_gathers_of_two_steps_in_flight_deliver_each_steps_fields(oracle):
    import torch.multiprocessing as mp
    ni, nj, world, nsteps = 24, 19, 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 977
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, ni, nj, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = oracle.synth_fields(ni, nj)
    assert sorted(got) == list(range(nsteps))
    for t in range(nsteps):
        ref = oracle.OracleSession("coare3p6", ni * nj, 1, False).compute(1, 2.0, 10.0, 2 + t, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"])
        np.testing.assert_array_equal(got[t], ref["ql"], err_msg=f"step {t}")


def test_committed_profile_is_of_this_device_code():
    """bench.py quotes the committed counter profile (HBM traffic, VALU share) only when it was taken with the very device code of this tree:
    the hash covers the flux kernels' translation unit, every header it includes and the compile flags.  A change to any of them must be
    followed by a re-take (tools/prof_quick.sh, tools/update_pmc.py --headline) before the round closes."""
    import json
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16
    pmc = json.load(open(bench.PMC_JSON))
    assert pmc["source_hash"] == h, f"{bench.PMC_JSON} was taken with other kernel sources: re-take it (tools/prof_quick.sh, tools/update_pmc.py --headline)"
    got = bench.committed_pmc("coare3p6", True, 4320, 3600, 5, "f64")
    assert got and not got.get("stale") and got["valu_insts_per_cell"] > 1000
    # a header of another translation unit (the helper kernels') does not enter the hash
    assert "ab_phymbl.hpp" not in open(os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_kernels.hip")).read()
