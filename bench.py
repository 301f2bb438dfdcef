#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X bulk-flux engine.

Metric (BASELINE.json): Mcell/s of one aerobulk_compute()-equivalent time record, COARE3p6 +
cool-skin/warm-layer on the 4320x3600 ORCA12 grid, fp64, inputs resident in HBM when the timed
region starts.  A "step" = one pass of the hot path over the whole grid (+ the RCCL gather of the
output fields to rank 0 when N > 1; the grid is j-block sharded across ranks: strong scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--algo coare3p6] [--no-skin] [--niter 5]
                    [--grid 4320x3600] [--precision f64] [--no-cpu-baseline]

For N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")


def algorithmic_bytes_per_cell(skin, esz):
    """no skin 6 in + 5 out; skin (single record) 8 in + 6 out."""
    return (8 + 6) * esz if skin else (6 + 5) * esz


def committed_pmc(algo, skin, ni, nj, niter, precision):
    """PMC figures of the committed rocprofv3 run for exactly this workload (profiles/r1_pmc.json), or None.
    bench.py cannot collect hardware counters itself; the numbers are measured by tools/prof_quick.sh on the same command."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc.json")) as fh:
            p = json.load(fh)
        c = p["config"]
        if (c["algo"], c["skin"], list(c["grid"]), c["nb_iter"], c["precision"]) == (algo, skin, [ni, nj], niter, precision):
            return p
    except Exception:
        pass
    return None


def fp64_flops(pmc):
    f = pmc["fp64_insts_per_launch"]
    return 64.0 * (2.0 * f["fma"] + f["mul"] + f["add"] + f["trans"])


def shard_rows(nj, world, rank):
    """Contiguous j-blocks: every rank owns ceil(nj/world) rows except possibly the last ones."""
    per = -(-nj // world)
    j0 = min(rank * per, nj)
    return j0, max(min(per, nj - j0), 0), per


def shard_rows_root_heavy(nj, world, rank, rows_peer):
    """j-blocks for the gathered run: ranks 1..N-1 own `rows_peer` rows each (behind rank 0's block), rank 0 owns the rest.
    Rank 0 is the destination of the gather: its own rows never cross a link, so it takes MORE rows than its peers until its
    compute time equals the time its peers need to compute AND ship theirs (balanced_peer_rows)."""
    rows_peer = max(1, min(int(rows_peer), nj // world if nj >= world else 1))
    nj0 = nj - (world - 1) * rows_peer
    if rank == 0:
        return 0, nj0, rows_peer
    return nj0 + (rank - 1) * rows_peer, rows_peer, rows_peer


def balanced_peer_rows(nj, world, t_cell, bytes_per_cell_on_link, link_bytes_per_s):
    """Rows per peer that equalise rank 0's compute, t_cell (1 - (N-1) f), with a peer's compute-and-ship time
    f max(t_cell, bytes/link_rate): f = t_cell / (P + (N-1) t_cell).  Fast links (P = t_cell) give the equal split 1/N."""
    p = max(t_cell, bytes_per_cell_on_link / max(link_bytes_per_s, 1.0))
    f = t_cell / (p + (world - 1) * t_cell)
    return max(1, min(int(round(f * nj)), max(nj // world, 1)))


def cpu_baseline(algo, skin, niter, zt, zu):
    """Reference Fortran (oracle/_ref, unmodified AeroBulk compiled with amdflang) timed on the host cores on a bounded
    sample of the same synthetic workload: first on ONE core (the reference is single-threaded), then on all cores at
    once (one independent process per core, j-block sharded like the GPU path).  Falls back to the C port if _ref did not
    travel."""
    import numpy as np
    from oracle import pyoracle as po
    ni, nj = 2160, 1440   # ~13 s of one-core reference work for the headline config
    n = ni * nj
    if not po.have_reference():
        f = po.synth_fields(ni, nj)
        s = po.OracleSession(algo, n, 1, skin)
        t0 = time.perf_counter()
        s.compute(1, zt, zu, niter, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"],
                  rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None)
        dt = time.perf_counter() - t0
        return {"value": round(n / dt / 1e6, 4), "unit": "Mcell/s", "cores": 1, "kind": "port",
                "sample": f"{algo}{'+skin' if skin else ''} nb_iter={niter} on a {ni}x{nj} slab of the same synthetic fields, {dt:.1f} s wall"}
    dt1 = po.run_reference_all_cores(algo, skin, niter, ni, nj, 1)[0]
    cores = max(1, min(os.cpu_count() or 1, 64))
    rows = 360                      # per process: 2160x360 cells, ~3.3 s of reference work, 90 MB
    secs = po.run_reference_all_cores(algo, skin, niter, ni, rows * cores, cores)
    v_all = ni * rows * cores / max(secs) / 1e6
    return {"value": round(v_all, 4), "unit": "Mcell/s", "cores": cores, "kind": "reference",
            "one_core_value": round(n / dt1 / 1e6, 4),
            "sample": f"{algo}{'+skin' if skin else ''} nb_iter={niter}, unmodified reference, one aerobulk_model(jt=1,Nt=1) call "
                      f"incl. AEROBULK_INIT per process: {cores} concurrent single-threaded processes x {ni}x{rows} cells of the "
                      f"same synthetic fields (slowest {max(secs):.1f} s); one process alone on {ni}x{nj}: {dt1:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--algo", default="coare3p6")
    ap.add_argument("--no-skin", action="store_true")
    ap.add_argument("--niter", type=int, default=5)
    ap.add_argument("--grid", default="4320x3600")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather (kernel-only scaling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 collective backend; gloo (outputs staged through host memory) exists only so that the sharded "
                         "path can be exercised on a box with fewer GPUs than ranks")
    ap.add_argument("--verify", action="store_true", help="N>1: rank 0 recomputes the whole grid alone and checks the gathered "
                                                        "fields are bit-identical (outside the timed region)")
    ap.add_argument("--gather-ts", action="store_true", help="N>1: also gather the skin temperature T_s (the north_star gather is "
                                                           "the output tau / Q_L / Q_H / E arrays: 5 fields; T_s stays on its GPU)")
    ap.add_argument("--peer-rows", type=int, default=0,
                    help="N>1 with the gather: rows owned by each of the ranks 1..N-1 (rank 0, the gather's destination, owns the "
                         "rest).  0 = measured: link rate and kernel rate are timed during set-up and the split balances rank 0's "
                         "compute against its peers' compute + transfer; -1 = equal split")
    ap.add_argument("--no-early-gather", action="store_true", help="N>1, RCCL: rank 0 joins the gather of chunk c after computing its "
                                                                    "own chunk c (instead of before)")
    ap.add_argument("--chunks", type=int, default=4, help="N>1: row sub-blocks per rank (gather of one overlaps compute of the next)")
    a = ap.parse_args()

    import torch
    import aerobulk_amd as ab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    ngpu = torch.cuda.device_count()
    dev_index = local_rank if a.backend == "nccl" else local_rank % max(ngpu, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    ni, nj = (int(x) for x in a.grid.lower().split("x"))
    skin = (not a.no_skin) and a.algo in ("coare3p0", "coare3p6", "ecmwf")
    zt, zu = 2.0, 10.0
    esz = 8 if a.precision == "f64" else 4
    tdt = torch.float64 if a.precision == "f64" else torch.float32
    nout = 6 if skin else 5
    names = ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")[:nout]
    ngat = nout if a.gather_ts else 5            # fields that travel to rank 0
    gathered = world > 1 and not a.no_gather

    # ---- sharding.  Without a gather: equal j-blocks.  With it: rank 0 (the destination) owns more rows than its peers
    # (shard_rows_root_heavy); how many is measured here unless --peer-rows says otherwise.
    tune = None
    if gathered:
        rows_peer = a.peer_rows if a.peer_rows > 0 else -(-nj // world)
        if a.peer_rows == 0:
            # (1) what one link delivers when all peers send at once: a few gathers of 32 MB per rank
            nb = 4 * 1024 * 1024
            tb = torch.zeros(nb, dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            tl = [torch.empty_like(tb) for _ in range(world)] if rank == 0 else None
            for _ in range(2):
                dist.gather(tb, tl, dst=0)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                dist.gather(tb, tl, dst=0)
            torch.cuda.synchronize()
            dist.barrier()
            t_gather = (time.perf_counter() - t0) / 5
            del tb, tl
            decision = torch.zeros(1, dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
            if rank == 0:
                try:
                    # (2) what one GPU computes: the same kernel on ~1 M cells of the same fields
                    rs = max(1, min(nj, (1 << 20) // ni))
                    fs = ab.synth_fields_device(ni, nj, 0, rs, precision=a.precision, device=dev, with_rad=True)
                    with ab.Session(a.algo, ni, rs, 1, skin, precision=a.precision, device=dev_index) as ss:
                        ss.set_humidity("sh")
                        best = 1e9
                        for it in range(40):
                            ss.compute(1, zt, zu, *[fs[k] for k in IN6], Niter=a.niter, rad_sw=fs["rad_sw"] if skin else None,
                                       rad_lw=fs["rad_lw"] if skin else None, want_T_s=skin, check=False)
                            if it >= 30:
                                best = min(best, ss.last_kernel_ms())
                    t_cell = best * 1e-3 / (ni * rs)
                    link = nb * 8 / t_gather
                    rows_peer = balanced_peer_rows(nj, world, t_cell, ngat * esz, link)
                    tune = {"link_GBps": round(link / 1e9, 1), "kernel_Mcell_per_s": round(1e-6 / t_cell, 1)}
                except Exception as e:      # never lose the run to the tuning: equal split
                    tune = {"failed": str(e)}
                decision[0] = rows_peer
            dist.broadcast(decision, src=0)
            rows_peer = int(decision.item())
        j0, njl, per = shard_rows_root_heavy(nj, world, rank, rows_peer)
        rows_peer = per
    else:
        j0, njl, per = shard_rows(nj, world, rank)
        rows_peer = per
    n_local = ni * njl

    # Each rank's j-block is cut into `chunks` row sub-blocks so that the RCCL gather of sub-block c overlaps the
    # kernel of sub-block c+1 (N == 1: one chunk, no communication).  Peers' chunks are padded to one common size (the
    # gather needs equal payloads); rank 0's own, larger block is cut into as many chunks of its own size.
    chunks = 1 if world == 1 else max(1, min(a.chunks, rows_peer))
    cr_peer = -(-rows_peer // chunks)            # rows per peer chunk (padded, identical on every peer)
    cr = -(-max(njl, 1) // chunks)               # rows per chunk of THIS rank
    n_cpad = ni * cr
    n_gpad = ni * cr_peer                        # cells per gathered chunk

    # synthetic inputs generated straight into HBM; outputs packed [chunk, field, cell] for ONE gather per chunk
    f = ab.synth_fields_device(ni, nj, j0, max(njl, 1), precision=a.precision, device=dev, with_rad=True)
    outbuf = torch.zeros((chunks, nout, n_cpad), dtype=tdt, device=dev)
    gather_lists = None
    send0 = None
    if gathered and rank == 0:     # rank 0's rows stay where they are; it joins the collective with an empty payload
        gather_lists = [[torch.empty((ngat, n_gpad), dtype=tdt, device=dev) for _ in range(world)] for _ in range(chunks)]
        send0 = torch.zeros((ngat, n_gpad), dtype=tdt, device=dev)

    work = []  # (session, inputs, rad, out) per non-empty chunk
    for c in range(chunks):
        r0 = min(c * cr, njl)
        rows = max(min(cr, njl - r0), 0)
        if rows == 0:
            work.append(None)
            continue
        lo, hi = r0 * ni, (r0 + rows) * ni
        sess = ab.Session(a.algo, ni, rows, 1, skin, precision=a.precision, device=dev_index)
        sess.set_humidity("sh")
        ins = [f[k][lo:hi] for k in IN6]
        rad = (f["rad_sw"][lo:hi], f["rad_lw"][lo:hi]) if skin else (None, None)
        out = {k: outbuf[c, i, :rows * ni] for i, k in enumerate(names)}
        work.append((sess, ins, rad, out))

    # Rank 0 contributes nothing to the gather (its rows stay where they are) but is the destination of every peer, and it owns
    # the largest block.  With RCCL it therefore joins the gather of chunk c BEFORE computing its own chunk c (the collective
    # then only waits for rank 0's chunk c-1): a peer's chunk lands while rank 0 is still computing, and the last gather of a
    # step overlaps rank 0's last chunk instead of following it.
    early = gathered and rank == 0 and a.backend == "nccl" and not a.no_early_gather

    def step():
        pending = []
        for c in range(chunks):
            if early:
                pending.append(dist.gather(send0, gather_lists[c], dst=0, async_op=True))
            w = work[c]
            if w is not None:
                sess, ins, rad, out = w
                sess.compute(1, zt, zu, *ins, Niter=a.niter, rad_sw=rad[0], rad_lw=rad[1], out=out, want_T_s=skin, check=False)
            if gathered and not early:
                payload = send0 if rank == 0 else outbuf[c, :ngat]
                if a.backend == "nccl":
                    pending.append(dist.gather(payload, gather_lists[c] if rank == 0 else None, dst=0, async_op=True))
                else:  # test-only path: gloo gathers host tensors
                    host = payload.cpu()
                    gl = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                    dist.gather(host, gl, dst=0)
                    if rank == 0:
                        for r in range(world):
                            gather_lists[c][r].copy_(gl[r])
        for p in pending:
            p.wait()      # stream-level wait (does not block the host)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def assemble():
        """rank 0: global fields [ngat, ni*nj]: its own block (never gathered) + the peers' gathered chunk buffers."""
        glob = torch.empty((ngat, ni * nj), dtype=tdt, device=dev)
        for c in range(chunks):                                   # own rows
            r0 = min(c * cr, njl)
            rows = max(min(cr, njl - r0), 0)
            if rows:
                glob[:, r0 * ni:(r0 + rows) * ni] = outbuf[c, :ngat, :rows * ni]
        for r in range(1, world):
            rj0, rnjl, _ = shard_rows_root_heavy(nj, world, r, rows_peer)
            for c in range(chunks):
                r0 = min(c * cr_peer, rnjl)
                rows = max(min(cr_peer, rnjl - r0), 0)
                if rows:
                    glob[:, (rj0 + r0) * ni:(rj0 + r0 + rows) * ni] = gather_lists[c][r][:, :rows * ni]
        return glob

    # pre-roll, part of the setup: the GPU raises its clocks during the first ~100 ms of sustained work (the first
    # configuration measured after start-up runs 5-8 % slow otherwise, profiles/r1_notes.md); then the W warm-up steps
    for _ in range(40):          # the same count on every rank (step() contains the gather)
        step()
    sync()
    for _ in range(a.warmup):
        step()
    sync()
    # timed region: EXACTLY K steps, no host sync inside
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for w in work:
        if w is not None:
            w[0].check()

    # per-launch kernel duration: HIP events recorded by the library around each launch, on the launch stream.
    # Reading an event pair synchronises, so this is a separate pass over the same inputs (not in `elapsed`).
    kdur = []
    for _ in range(min(a.steps, 10)):
        tot = 0.0
        for w in work:
            if w is not None:
                sess, ins, rad, out = w
                sess.compute(1, zt, zu, *ins, Niter=a.niter, rad_sw=rad[0], rad_lw=rad[1], out=out, want_T_s=skin, check=False)
                tot += sess.last_kernel_ms()
        kdur.append(tot)
    k_ms = sum(kdur) / max(len(kdur), 1)

    verify_msg = None
    if a.verify and gathered and rank == 0:
        glob = assemble()
        ff = ab.synth_fields_device(ni, nj, precision=a.precision, device=dev, with_rad=True)
        with ab.Session(a.algo, ni, nj, 1, skin, precision=a.precision, device=dev_index) as s1:
            s1.set_humidity("sh")
            one = s1.compute(1, zt, zu, *[ff[k] for k in IN6], Niter=a.niter, rad_sw=ff["rad_sw"] if skin else None,
                             rad_lw=ff["rad_lw"] if skin else None, want_T_s=skin)
        bad = [k for i, k in enumerate(names[:ngat]) if not torch.equal(glob[i], one[k])]
        verify_msg = "gathered == single-GPU (bit-identical)" if not bad else f"MISMATCH in {bad}"
        if bad:
            raise SystemExit("verify failed: " + verify_msg)

    if rank == 0:
        cells = ni * nj
        ms_per_step = elapsed / a.steps * 1e3
        value = cells * a.steps / elapsed / 1e6
        bpc = algorithmic_bytes_per_cell(skin, esz)
        achieved = bpc * n_local / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pmc = committed_pmc(a.algo, skin, ni, nj, a.niter, a.precision) if world == 1 else None
        res = {
            "metric": "Mcell/s COARE3p6+cool-skin on 4320x3600 grid" if (a.algo == "coare3p6" and skin and (ni, nj) == (4320, 3600))
                      else f"Mcell/s {a.algo}{'+skin' if skin else ''} on {ni}x{nj} grid",
            "value": round(value, 2), "unit": "Mcell/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.algo}{' + cool-skin/warm-layer' if skin else ''}, {ni}x{nj} grid, nb_iter={a.niter}, "
                                   f"zt=2 zu=10, one time record (jt=1=Nt), inputs/outputs resident in HBM",
                       "grid": [ni, nj], "algo": a.algo, "skin": skin, "nb_iter": a.niter,
                       "sharding": f"j-block x{world}" + ("" if not gathered else
                                                            f": rank 0 owns {nj - (world - 1) * rows_peer} rows, ranks 1..{world - 1} {rows_peer} rows each "
                                                            f"(rank 0 is the gather's destination; its compute is balanced against the peers' compute + transfer)"
                                                            f" + RCCL gather of {', '.join(names[:ngat])} to rank 0, {chunks} overlapped row chunks per rank"),
                       **({"sharding_tuning": tune} if tune else {})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": round(pmc["traffic_bytes_per_launch"]) if pmc else None,
                         "kernel": f"flux_kernel<{a.precision},{a.algo},{'skin' if skin else 'noskin'}>",
                         "kernel_ms": round(k_ms, 4), "bytes_per_cell": bpc, "cells_per_launch": n_local,
                         "binding_resource": "fp64 VALU issue" if a.precision == "f64" else "fp32 VALU issue",
                         "valu_busy": round(pmc["valu_busy"], 3) if pmc else None,
                         "valu_insts_per_cell": round(pmc["valu_insts_per_cell"]) if pmc else None,
                         # fp64 arithmetic actually issued (PMC wave-instruction counts x 64 lanes, an FMA = 2 flops) over the
                         # live kernel time, against the fp64 vector peak (MI355X_MICROARCH.md: 78.6 TFLOP/s)
                         "fp64_tflops": round(fp64_flops(pmc) / (k_ms * 1e-3) / 1e12, 2) if (pmc and k_ms > 0) else None,
                         "fp64_vector_peak_tflops": 78.6,
                         "note": "the kernel is VALU-bound (hundreds of fp64 transcendentals per cell), not HBM-bound; traffic/valu_* are "
                                 "rocprofv3 PMC figures of the committed profile of this same command (profiles/), DESIGN.md §3.1"},
        }
        if verify_msg:
            res["verify"] = verify_msg
        if not a.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(a.algo, skin, a.niter, zt, zu)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "Mcell/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res), flush=True)
    for w in work:
        if w is not None:
            w[0].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
