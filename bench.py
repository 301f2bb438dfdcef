#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X bulk-flux engine.

Metric (BASELINE.json): Mcell/s of one aerobulk_compute()-equivalent time record, COARE3p6 +
cool-skin/warm-layer on the 4320x3600 ORCA12 grid, fp64, inputs resident in HBM when the timed
region starts.  A "step" = one pass of the hot path over the whole grid (+ the RCCL gather of the
output fields to rank 0 when N > 1; the grid is j-block sharded across ranks: strong scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--algo ..] [--no-skin] [--niter 5]
                    [--grid 4320x3600] [--precision f64] [--resident] [--no-cpu-baseline]

--config picks a BASELINE.json configuration (default 3 = the headline):
   1  NCAR, no skin, 360x180, nb_iter=5: the reference's own CPU-runnable case; the GPU runs it (launch-bound: 65 k cells) and
      the cpu_baseline leg times the unmodified reference on exactly this workload (the product itself has no CPU path)
   2  COARE3p6, no skin, 1440x1080, nb_iter=8
   3  COARE3p6 + cool-skin/warm-layer, 4320x3600, nb_iter=5
   4  all five algorithms back-to-back on 4320x3600 (no skin scheme for any: one consistent setting; --skin adds
      the scheme for the three algorithms that have one); a step = the five passes; value = 5 x cells per step
   5  ECMWF + cool-skin/warm-layer, fp32 path, 12960x10800
N > 1: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU; the driver's form), or plain
`python bench.py --gpus N` = ONE process with one sharded library session over N devices and the library's own in-process RCCL
gather (main_inprocess; --launcher torchrun starts the former as a child instead).  ONE JSON line is printed.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TF = 78.6   # fp64 vector (VALU) peak, same guide: half the 157.3 TFLOP/s fp32 vector rate
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
SKIN_ALGOS = ("coare3p0", "coare3p6", "ecmwf")
ALL_ALGOS = ("coare3p0", "coare3p6", "ncar", "ecmwf", "andreas")
PMC_JSON = os.path.join(ROOT, "profiles", "r6_pmc.json")      # this round's counter profile of the headline kernel (tools/update_pmc.py --headline)


def kernel_label(precision, algo, skin, n_cells, ab):
    """Name of the kernel a launch of n_cells runs (launch_t, ab_kernels.hip): fp64 COARE with the skin schemes on grids of three tiles per
    team and more takes flux_kernel_cu (one workgroup of sixteen waves per CU, every table in LDS); everything else the 256-thread flux_kernel."""
    cu = (precision in ("f64", "f32_storage") and algo in ("coare3p0", "coare3p6") and skin and os.environ.get("AEROBULK_AMD_CU_KERNEL", "")[:1] != "0")
    if cu:
        import torch
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        tiles = (n_cells - min(n_cells, cus * 4 * 256)) // 512 + -(-min(n_cells, cus * 4 * 256) // 256)
        cu = tiles >= 3 * cus * 4
    return f"{'flux_kernel_cu' if cu else 'flux_kernel'}<{precision},{algo},{'skin' if skin else 'noskin'}>"


def algorithmic_bytes_per_cell(skin, esz):
    """no skin 6 in + 5 out; skin (single record) 8 in + 6 out."""
    return (8 + 6) * esz if skin else (6 + 5) * esz


def kernel_source_hash():
    """Identity of the device code the committed PMC profile was taken with: the flux kernels' translation unit (ab_kernels.hip), every
    header of this repository it includes (the closure of its `#include "..."` lines, so that a change to a header of ANOTHER translation
    unit — the helper kernels of ab_phymbl.hip — does not disown the profile) and the compile flags."""
    import re
    from aerobulk_amd import build as b
    csrc = os.path.join(ROOT, "aerobulk_amd", "csrc")
    seen, todo = [], ["ab_kernels.hip"]
    while todo:
        name = todo.pop()
        path = os.path.normpath(os.path.join(csrc, name))
        if path in seen or not os.path.exists(path):
            continue
        seen.append(path)
        with open(path, "r", errors="replace") as fh:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
                todo.append(os.path.join(os.path.dirname(os.path.relpath(path, csrc)), inc))
    h = hashlib.sha256()
    for path in sorted(seen):
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(b.HIPFLAGS).encode())
    return h.hexdigest()[:16]


def committed_pmc(algo, skin, ni, nj, niter, precision):
    """Hardware-counter figures of the committed rocprofv3 run of exactly this workload (profiles/r6_pmc.json, written by
    tools/update_pmc.py from a tools/prof_quick.sh run), or None.  bench.py cannot collect counters itself.  They are only
    quoted when the profile was taken with the device code that is running now (source hash), and every figure derived from
    them uses the PROFILE's own kernel duration, never a live timing."""
    try:
        with open(PMC_JSON) as fh:
            p = json.load(fh)
        c = p["config"]
        if (c["algo"], c["skin"], list(c["grid"]), c["nb_iter"], c["precision"]) != (algo, skin, [ni, nj], niter, precision):
            return None
        if p.get("source_hash") != kernel_source_hash():
            return {"stale": True, "source": p.get("source")}
        return p
    except Exception:
        return None


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


# top level of every JSON line, beside roofline.bound = "hbm" (the contract's vocabulary is hbm | mfma; BASELINE.json asks for the HBM fraction):
# no reader should take a 90 %-VALU-busy kernel for a memory-bound one
LIMITER_NOTE = ("valu_fp64: the flux kernels are bound by fp64 vector-ALU issue (hundreds of transcendentals per cell, ~90 % VALU busy in "
                "rocprofv3), NOT by HBM and not by MFMA (pointwise, no contraction); roofline.frac is the HBM fraction BASELINE.json asks for, "
                "roofline.fp64_frac / valu_issue_frac the binding resource")
# What ab_calibrate's fp64 FMA workload delivered on the leases of round 6 (65.8 ... 66.05 TFLOP/s at a reported sclk of 2.37-2.40 GHz: 84 % of the
# guide's 78.6 — the chains do not reach the peak issue rate, which is immaterial: the workload is FIXED, only its ratio between boxes is used).
# value_norm = value x CALIB_REF_TFLOPS / measured: `value` as a box of that speed would have delivered it.
CALIB_REF_TFLOPS = 66.0


def read_sclk_mhz(dev_index):
    """Shader clock the driver reports for the device right now (sysfs pp_dpm_sclk, the level marked '*'), or None.  Read without
    starting a process; informational (calibrate_box samples it while its workload runs and keeps the highest reading) — the calibration
    kernel's own rate is the figure that is compared."""
    import glob
    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        path = cards[dev_index] if dev_index < len(cards) else cards[0]
        with open(path) as fh:
            for ln in fh:
                if "*" in ln:
                    return int("".join(c for c in ln.split(":")[1] if c.isdigit()))
    except Exception:
        pass
    return None


def calibrate_box(ab, dev_index, launches):
    """The box, not the kernel: `launches` runs of ab_calibrate's fixed fp64 FMA workload (about 1.5 ms each; the first ones also
    raise the clocks), median of the last five.  Called before the pre-roll and again after the timed region."""
    try:
        import threading
        samples, stop = [], threading.Event()

        def sampler():      # the driver's shader clock WHILE the workload runs (read after it, the device is back at its idle level: 383 MHz)
            while not stop.is_set():
                v = read_sclk_mhz(dev_index)
                if v:
                    samples.append(v)
                time.sleep(0.002)
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        try:
            tf = [ab.calibrate("fma_f64", dev_index)[1] for _ in range(launches)]      # (ctypes releases the GIL during the call)
        finally:
            stop.set()
            th.join(timeout=1.0)
        tail = sorted(tf[-5:])
        rec = {"fma_f64_tflops": round(tail[len(tail) // 2], 3), "sclk_mhz": max(samples) if samples else None}
        if launches >= 10:          # once per run, before the pre-roll: the box's memory side (1 GiB device-to-device copy, read + written)
            try:
                rec["hbm_copy_GBps"] = round(sorted(ab.calibrate("hbm_copy", dev_index)[1] for _ in range(3))[1], 1)
            except Exception:
                rec["hbm_copy_GBps"] = None
        return rec
    except Exception as e:          # a report, never a reason to lose the run
        return {"fma_f64_tflops": None, "error": str(e)}


def calib_record(before, after, value):
    """`calib` of the JSON line + value_norm = value x (CALIB_REF_TFLOPS / the box's measured fp64 FMA rate): what the same kernel
    would have delivered on a box as fast as the leases the reference rate was taken on.  Compare ROUNDS by value_norm, boxes by calib (README.md).
    What it does NOT remove: two round-6 leases 2.5 % apart in this rate ran the headline kernel 0.5 % apart the OTHER way (value_norm 8 072 and 7 849) —
    the flux kernel's mix (LDS, quarter-rate seeds, 30 % of its cycles waiting) answers a box's power management differently from pure FMA chains.  The
    calibration bounds a box's contribution to a few per cent; the instruction count per cell (profiles/r6_pmc.json) is the clock-free figure."""
    rates = [c["fma_f64_tflops"] for c in (before, after) if c and c.get("fma_f64_tflops")]
    rec = {"fma_f64_tflops_before": before.get("fma_f64_tflops") if before else None,
           "fma_f64_tflops_after": after.get("fma_f64_tflops") if after else None,
           "sclk_mhz": (after or {}).get("sclk_mhz") or (before or {}).get("sclk_mhz"),
           "hbm_copy_GBps": (before or {}).get("hbm_copy_GBps"),
           "reference_tflops": CALIB_REF_TFLOPS,
           "workload": "ab_calibrate(AB_CALIB_FMA_F64): chains of v_fma_f64, 4 waves per SIMD, ~1.5 ms, HIP events; median of 5 launches "
                       "before the pre-roll (after 15 ramp launches) and after the timed region"}
    norm = None
    if rates:
        mean = sum(rates) / len(rates)
        rec["box_speed"] = round(mean / CALIB_REF_TFLOPS, 4)            # 1.0 = a round-6 lease; the factor `value` was divided by
        norm = round(value * CALIB_REF_TFLOPS / mean, 2)
    return rec, norm


def gather_model(n_gpus, cells, t_cell_slab_s, one_gpu_ms, bytes_per_cell_on_link, link_rates_GBps):
    """What a gathered N-GPU step should cost, from three stated inputs — so that the first run on N real devices confirms or kills it.
    Root-heavy cut (balanced_peer_rows): a peer owns the fraction f of the cells, f = t / (p + (N-1) t) with t = kernel time per cell at
    slab size and p = max(t, bytes on the link per cell / link rate): a peer's compute + ship of its slab takes as long as the root's
    compute of the rest.  step = t x cells x (1 - (N-1) f); speed-up = one-GPU step / that."""
    pred = {}
    for L in link_rates_GBps:
        p_ = max(t_cell_slab_s, bytes_per_cell_on_link / (L * 1e9))
        f = t_cell_slab_s / (p_ + (n_gpus - 1) * t_cell_slab_s)
        step_ms = t_cell_slab_s * cells * (1.0 - (n_gpus - 1) * f) * 1e3
        pred[f"{L:g} GB/s per peer"] = {"peer_share": round(f, 4), "root_share": round(1.0 - (n_gpus - 1) * f, 4), "step_ms": round(step_ms, 4),
                                        "speedup_vs_one_gpu": round(one_gpu_ms / step_ms, 2) if step_ms > 0 else None}
    return {"inputs": {"n_gpus": n_gpus, "cells": cells, "kernel_ns_per_cell_at_slab_size": round(t_cell_slab_s * 1e9, 5),
                       "one_gpu_step_ms": round(one_gpu_ms, 4), "bytes_per_cell_on_link": bytes_per_cell_on_link},
            "formula": "f = t/(max(t, b/L) + (N-1) t); step = t cells (1 - (N-1) f); speedup = one_gpu_step / step",
            "predicted": pred,
            "gather_free_ceiling": round(one_gpu_ms / (t_cell_slab_s * cells / n_gpus * 1e3), 2),
            "note": "xGMI: 7 links x ~153 GB/s peak per GPU (MI355X guide); what one peer sustains into the root while six others send is the "
                    "number the first 8-GPU run measures (split_tuning.link_GBps_per_peer) — rates above are assumptions, not measurements"}


def host_path_leg(ab, sess, fields_dev, skin, zt, zu, niter, n, np_dtype, sync, reduce_max, records=4):
    """The reference's own calling convention at the GCM call site (mod_aerobulk.f90:246-262): caller arrays in PAGEABLE HOST memory, every
    record H2D | kernel | D2H through the session's AB_MEM_HOST entry (chunk-pipelined staging; a sharded session cuts the arrays by rows and
    every device moves its rows over its own PCIe link).  Outside the timed region; never `value`.  first_record_ms includes the staging
    buffers' allocation."""
    import numpy as np
    names = IN6 + (("rad_sw", "rad_lw") if skin else ())
    h = {k: np.ascontiguousarray(fields_dev[k].cpu().numpy()) for k in names}
    out = {k: np.zeros(n, dtype=np_dtype) for k in (("QL", "QH", "Tau_x", "Tau_y", "Evap") + (("T_s",) if skin else ()))}    # caller-owned, touched
    ts = []
    for _ in range(records):
        sync()
        t0 = time.perf_counter()
        sess.compute(1, zt, zu, *[h[k] for k in IN6], Niter=niter, rad_sw=h.get("rad_sw"), rad_lw=h.get("rad_lw"), out=out, want_T_s=skin)
        ts.append(reduce_max(time.perf_counter() - t0))
    steady = min(ts[1:])
    return {"first_record_ms": round(ts[0] * 1e3, 3), "ms_per_record": round(steady * 1e3, 3), "records": records,
            "bytes_per_cell_over_pcie": (len(names) + len(out)) * np.dtype(np_dtype).itemsize, "checksum_QL": float(out["QL"].sum(dtype=np.float64)),
            "what": "ab_session_compute(AB_MEM_HOST) on pageable caller arrays, one record = H2D of the inputs + kernel + D2H of the fluxes, "
                    "best of the records after the first; the reference's AEROBULK_MODEL call site"}


def overlapped_leg(ab, torch, passes, ni, rows, precision, dev_index, dev, f, names, tdt, niter, zt, zu, steps, sync, reduce_max):
    """Consecutive records that do NOT depend on one another (single-record sessions: jt = 1 = Nt, as this benchmark's; ensemble members, regions),
    issued alternately on TWO streams through TWO sessions: the persistent workgroups of record t + 1 take each CU the moment record t's
    workgroup leaves it, which removes the start and the drain of a launch from the time per record (profiles/r6_notes.md §10).  A time loop
    that carries the warm layer's state cannot do this; reported beside `value` / `resident`, never as them.  Returns seconds per step (max over
    ranks) or None."""
    n = ni * rows
    if n <= 0:                # a rank without rows joins the other ranks' barriers and reduction
        sync()
        sync()
        reduce_max(0.0)
        return None
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    sets = []
    for _ in range(2):
        per = []
        for algo, skin in passes:
            sess = ab.Session(algo, ni, rows, 1, skin, precision=precision, device=dev_index)
            sess.set_humidity("sh")
            out = {k: torch.empty(n, dtype=tdt, device=dev) for k in names if (k != "T_s" or skin)}
            per.append((sess, skin, out))
        sets.append(per)

    def step(i):
        with torch.cuda.stream(streams[i & 1]):
            for sess, skin, out in sets[i & 1]:
                sess.compute(1, zt, zu, *[f[k][:n] for k in IN6], Niter=niter, rad_sw=f["rad_sw"][:n] if skin else None,
                             rad_lw=f["rad_lw"][:n] if skin else None, out=out, want_T_s=skin, check=False)
    for i in range(6):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    el = reduce_max(time.perf_counter() - t0)
    for per in sets:
        for sess, _, _ in per:
            sess.check()
            sess.close()
    return el / steps


def cpu_baseline_config1(niter, zt, zu):
    """BASELINE config 1 on the CPU, as the reference runs it: NCAR, 360x180, one aerobulk_model(jt=1,Nt=1) call incl.
    AEROBULK_INIT, the unmodified reference (oracle/_ref) in one process; the C port if _ref did not travel."""
    from oracle import pyoracle as po
    ni, nj = 360, 180
    if po.have_reference():
        dt = min(po.run_reference_all_cores("ncar", False, niter, ni, nj, 1)[0] for _ in range(3))
        kind = "reference"
    else:
        f = po.synth_fields(ni, nj)
        s = po.OracleSession("ncar", ni * nj, 1, False)
        t0 = time.perf_counter()
        s.compute(1, zt, zu, niter, f["sst"], f["t_zt"], f["hum_zt"], f["u_zu"], f["v_zu"], f["slp"])
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": round(ni * nj / dt / 1e6, 4), "unit": "Mcell/s", "cores": 1, "kind": kind,
            "sample": f"the whole of config 1: ncar nb_iter={niter} on the 360x180 synthetic grid, one aerobulk_model(jt=1,Nt=1) call "
                      f"incl. AEROBULK_INIT, single process, {dt * 1e3:.1f} ms"}


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
    # one process per PHYSICAL core where the topology can be read (SMT siblings share the fp64 units: two processes per core gave
    # 13.5 x one core on 64 logical CPUs in round 2), each pinned to its own core; capped so that the sample stays bounded
    logical = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, physical = set(), []
    for c in logical:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                key = fh.read().strip()
        except OSError:
            key = str(c)
        if key not in seen:
            seen.add(key)
            physical.append(c)
    cap = 64
    cpus = physical[:cap]
    cores = len(cpus)
    dt1 = po.run_reference_all_cores(algo, skin, niter, ni, nj, 1, cpus=cpus[:1])[0]
    rows = 360                      # per process: 2160x360 cells, ~3.3 s of reference work, 90 MB
    secs = po.run_reference_all_cores(algo, skin, niter, ni, rows * cores, cores, cpus=cpus)
    v_all = ni * rows * cores / max(secs) / 1e6
    return {"value": round(v_all, 4), "unit": "Mcell/s", "cores": cores, "kind": "reference",
            "one_core_value": round(n / dt1 / 1e6, 4),
            "host": {"logical_cpus": os.cpu_count(), "usable_logical_cpus": len(logical), "physical_cores": len(physical), "process_cap": cap},
            "sample": f"{algo}{'+skin' if skin else ''} nb_iter={niter}, unmodified reference, one aerobulk_model(jt=1,Nt=1) call "
                      f"incl. AEROBULK_INIT per process: {cores} concurrent single-threaded processes, each pinned to its own physical core "
                      f"(host: {os.cpu_count()} logical CPUs, {len(physical)} physical cores; cap {cap} processes), x {ni}x{rows} cells of the "
                      f"same synthetic fields (slowest {max(secs):.1f} s); one process alone on {ni}x{nj}: {dt1:.1f} s"}


def resolve_config(a):
    """(passes, grid, precision, niter, label): passes = [(algo, skin)] run back-to-back in one step."""
    explicit_grid = a.grid is not None
    if a.config == 1:
        passes, grid, prec, niter = [("ncar", False)], "360x180", "f64", 5
    elif a.config == 2:
        passes, grid, prec, niter = [("coare3p6", False)], "1440x1080", "f64", 8
    elif a.config == 4:
        passes = [(al, bool(a.skin) and al in SKIN_ALGOS) for al in ALL_ALGOS]
        grid, prec, niter = "4320x3600", "f64", 5
    elif a.config == 5:
        passes, grid, prec, niter = [("ecmwf", True)], "12960x10800", "f32_mixed", 5
    else:
        skin = (not a.no_skin) and a.algo in SKIN_ALGOS
        passes, grid, prec, niter = [(a.algo, skin)], "4320x3600", "f64", 5
    if explicit_grid:
        grid = a.grid
    if a.precision is not None:
        prec = a.precision
    if a.niter is not None:
        niter = a.niter
    return passes, grid, prec, niter


def main_inprocess(a):
    """`python bench.py --gpus N` as typed: ONE process, one sharded library session per pass over N devices.  Every shard's fields
    are resident on its device; a step = ab_session_compute_shards (kernels enqueued on every device) + ab_session_gather of
    tau / Q_L / Q_H / E (+ T_s with --gather-ts) to the device of shard 0 by RCCL send / recv inside the process
    (ncclCommInitAll) — the north_star's layout behind AEROBULK_MODEL's call site (mod_aerobulk.f90:250-262).  Two sets of output /
    destination buffers: the gather of step t (on communication streams) drains while step t + 1 computes."""
    import torch
    import aerobulk_amd as ab

    nsh = a.gpus
    devs = [int(x) for x in a.devices.split(",")] if a.devices else list(range(nsh))
    if len(devs) != nsh:
        raise SystemExit(f"--devices lists {len(devs)} shards, --gpus says {nsh}")
    nvis = torch.cuda.device_count()
    if max(devs) >= nvis:
        raise SystemExit(f"--gpus {nsh}: device {max(devs)} asked for, {nvis} visible (one-GPU box: --devices 0,0,... puts the shards on device 0)")
    passes, grid, precision, niter = resolve_config(a)
    ni, nj = (int(x) for x in grid.lower().split("x"))
    npass = len(passes)
    any_skin = any(sk for _, sk in passes)
    zt, zu = 2.0, 10.0
    esz = 8 if precision == "f64" else 4
    tdt = torch.float64 if precision == "f64" else torch.float32
    dtype_label = {"f64": "f64", "f32": "f32", "f32_storage": "f32 arrays / f64 arithmetic",
                   "f32_mixed": "f32 arrays / f64 anchors + f32 transcendentals (AB_F32_MIXED)"}[precision]
    gnames = ("QL", "QH", "Tau_x", "Tau_y", "Evap") + (("T_s",) if (a.gather_ts and any_skin) else ())
    root = 0
    root_dev = torch.device("cuda", devs[root])
    torch.cuda.set_device(root_dev)
    nbuf = 1 if a.no_pipeline_gather else 2

    class Setup:
        """Sessions, resident fields and buffers for one cut of the grid into row blocks (`rows`: per shard, None = equal blocks)."""
        pass

    def setup(rows):
        st = Setup()
        sessions = [ab.Session(algo, ni, nj, 1, skin, precision=precision, device=devs, rows=rows) for algo, skin in passes]
        for s_ in sessions:
            s_.set_humidity("sh")
        shards = sessions[0].shards()                      # [(j0, njl, device)]: the same cut for every pass
        fields, cstream, mstream = [], [], []
        for j0, njl, d in shards:
            with torch.cuda.device(d):
                fields.append(ab.synth_fields_device(ni, nj, j0, njl, precision=precision, device=torch.device("cuda", d), with_rad=True))
                cstream.append(torch.cuda.Stream(device=d))
                mstream.append(torch.cuda.Stream(device=d))
        torch.cuda.synchronize()
        # destinations on the root's device: [buffer set][pass] -> dict of whole-grid tensors; the root shard computes straight into its rows
        dst = [[{k: torch.zeros(ni * nj, dtype=tdt, device=root_dev) for k in gnames} for _ in range(npass)] for _ in range(nbuf)]
        outs = []                                          # [buffer set][pass][shard] -> dict of that shard's output tensors
        for b in range(nbuf):
            per_pass = []
            for p, (algo, skin) in enumerate(passes):
                per_shard = []
                for r, (j0, njl, d) in enumerate(shards):
                    names = ("QL", "QH", "Tau_x", "Tau_y", "Evap") + (("T_s",) if skin else ())
                    if r == root:
                        o = {k: (dst[b][p][k][j0 * ni:(j0 + njl) * ni] if k in dst[b][p] else torch.empty(ni * njl, dtype=tdt, device=root_dev)) for k in names}
                    else:
                        o = {k: torch.empty(ni * njl, dtype=tdt, device=torch.device("cuda", d)) for k in names}
                    per_shard.append(o)
                per_pass.append(per_shard)
            outs.append(per_pass)
        shard_in = [[{k: f[k] for k in (IN6 + (("rad_sw", "rad_lw") if skin else ()))} for f in fields] for _, skin in passes]
        c_ptr = [st.cuda_stream for st in cstream]
        m_ptr = [st.cuda_stream for st in mstream]
        gdone = [[None] * len(shards) for _ in range(nbuf)]      # per buffer set and shard: event "the gathers that read this set are done"
        nstep = [0]

        def step(with_gather=True, nit=None):
            b = nstep[0] % nbuf
            nstep[0] += 1
            for r, (_, _, d) in enumerate(shards):               # the set is written again only after its gathers of nbuf steps ago
                if gdone[b][r] is not None:
                    cstream[r].wait_event(gdone[b][r])
            for p, sess in enumerate(sessions):
                sess.compute_shards(1, zt, zu, shard_in[p], outs[b][p], Niter=niter if nit is None else nit, streams=c_ptr, check=False)
                if with_gather:
                    for r in range(len(shards)):                 # communication streams: behind the kernels of this pass
                        mstream[r].wait_stream(cstream[r])
                    sess.gather([{k: v for k, v in o.items() if k in gnames} for o in outs[b][p]], dst[b][p], root=root, streams=m_ptr, synchronize=False)
            if with_gather:
                for r, (_, _, d) in enumerate(shards):
                    ev = torch.cuda.Event()
                    ev.record(mstream[r])
                    gdone[b][r] = ev
                # the root's compute stream also waits for what was received into this set before writing it again
                ev = torch.cuda.Event()
                ev.record(mstream[root])
                gdone[b][root] = ev

        def sync():
            for d in sorted(set(devs)):
                torch.cuda.synchronize(d)
        st.sessions, st.shards, st.step, st.sync, st.cstream, st.c_ptr, st.shard_in, st.outs, st.dst, st.nstep = sessions, shards, step, sync, cstream, c_ptr, shard_in, outs, dst, nstep
        return st

    def kernel_ms(st, nrep):
        """per device: events on every shard's compute stream around the launches of ALL passes of one step"""
        kms = [0.0] * len(st.shards)
        for _ in range(nrep):
            evs = []
            for r in range(len(st.shards)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st.cstream[r])
                evs.append((e0, e1))
            for p, sess in enumerate(st.sessions):
                sess.compute_shards(1, zt, zu, st.shard_in[p], st.outs[0][p], Niter=niter, streams=st.c_ptr, check=False)
            for r in range(len(st.shards)):
                evs[r][1].record(st.cstream[r])
            st.sync()
            for r in range(len(st.shards)):
                kms[r] += evs[r][0].elapsed_time(evs[r][1]) / nrep
        return kms

    # ---- the cut.  Shard 0 lives on the gather's destination: its rows never cross a link, so it is given MORE rows than its peers
    # until its kernels take as long as a peer's kernels + that peer's share of the root's ingress (balanced_peer_rows, as the
    # torchrun path).  Measured on the equal cut first: per-device kernel time and what a gather of these fields costs.
    tune = None
    rows = None
    if nsh > 1 and a.peer_rows >= 0 and nj >= 2 * nsh:
        if a.peer_rows > 0:
            rp = max(1, min(a.peer_rows, nj // nsh))
        else:
            st = setup(None)
            for _ in range(6):
                st.step()
            st.sync()
            kms0 = kernel_ms(st, 4)
            t_cell = max(kms0[r] * 1e-3 / (ni * st.shards[r][1]) for r in range(nsh))       # all passes, per cell
            t0 = time.perf_counter()
            for _ in range(4):
                st.step(True)
            st.sync()
            t_g = (time.perf_counter() - t0) / 4
            t0 = time.perf_counter()
            for _ in range(4):
                st.step(False)
            st.sync()
            t_c = (time.perf_counter() - t0) / 4
            # ingress of the root per step beyond what the kernels hide: a lower bound of the link time of one peer's payload
            peer_bytes = npass * len(gnames) * esz * ni * st.shards[1][1]
            t_link = max(t_g - t_c, 1e-6)
            link = peer_bytes * (nsh - 1) / t_link / max(nsh - 1, 1)                           # per peer, all sending at once
            rp = balanced_peer_rows(nj, nsh, t_cell, npass * len(gnames) * esz, link)
            tune = {"equal_cut_step_ms": round(t_g * 1e3, 4), "equal_cut_resident_step_ms": round(t_c * 1e3, 4),
                    "link_GBps_per_peer": round(link / 1e9, 1), "kernel_Mcell_per_s_per_device": round(npass * 1e-6 / t_cell, 1)}
            for s_ in st.sessions:
                s_.close()
            del st
            torch.cuda.empty_cache()
        rows = [nj - (nsh - 1) * rp] + [rp] * (nsh - 1)
    st = setup(rows)
    sessions, shards, step, sync, cstream, c_ptr, shard_in, outs, dst, nstep = st.sessions, st.shards, st.step, st.sync, st.cstream, st.c_ptr, st.shard_in, st.outs, st.dst, st.nstep

    def timed(nsteps, with_gather):
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(with_gather)
        sync()
        return time.perf_counter() - t0

    calib_before = calibrate_box(ab, devs[root], 20)
    for _ in range(max(4, 40 // npass) if ni * nj <= 4320 * 3600 else 4):     # clock ramp, as the other path
        step()
    sync()
    for _ in range(a.warmup):
        step()
    elapsed = timed(a.steps, True)
    calib_after = calibrate_box(ab, devs[root], 5)
    elapsed_resident = timed(a.steps, False)
    for s_ in sessions:
        s_.check()

    # per-device kernel duration of the headline pass alone (the roofline's kernel) and of all passes of a step
    st1 = Setup()
    st1.sessions, st1.shards, st1.cstream, st1.c_ptr, st1.shard_in, st1.outs, st1.sync = sessions[:1], shards, cstream, c_ptr, shard_in[:1], [o[:1] for o in outs], sync
    kms = kernel_ms(st1, min(a.steps, 10))
    kms_all = kms if npass == 1 else kernel_ms(st, min(a.steps, 10))

    # the one-GPU step of the same workload on the root's device, in this very run: numerator of every speed-up quoted below
    one_gpu_ms = None
    host_path = None
    peer_slab = None
    try:
        ffull = ab.synth_fields_device(ni, nj, precision=precision, device=root_dev, with_rad=True)
        o1 = {k: torch.empty(ni * nj, dtype=tdt, device=root_dev) for k in ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")}
        one_gpu_ms = 0.0
        for algo, skin in passes:
            with ab.Session(algo, ni, nj, 1, skin, precision=precision, device=devs[root]) as s1:
                s1.set_humidity("sh")
                kw = dict(Niter=niter, rad_sw=ffull["rad_sw"] if skin else None, rad_lw=ffull["rad_lw"] if skin else None,
                          out={k: v for k, v in o1.items() if (k != "T_s" or skin)}, want_T_s=skin, check=False)
                for _ in range(3):
                    s1.compute(1, zt, zu, *[ffull[k] for k in IN6], **kw)
                torch.cuda.synchronize(root_dev)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    s1.compute(1, zt, zu, *[ffull[k] for k in IN6], **kw)
                torch.cuda.synchronize(root_dev)
                one_gpu_ms += (time.perf_counter() - t0) / a.steps * 1e3
        # ... and a PEER's slab alone on the same device (the model's kernel time per cell at slab size: with several shards on one device —
        # the one-GPU box — the per-shard timings above overlap and say nothing about a device that owns one slab)
        if nsh > 1:
            rows_peer_ = shards[1][1]
            n_peer = ni * rows_peer_
            slab_ms = 0.0
            for algo, skin in passes:
                with ab.Session(algo, ni, rows_peer_, 1, skin, precision=precision, device=devs[root]) as s1:
                    s1.set_humidity("sh")
                    kw = dict(Niter=niter, rad_sw=ffull["rad_sw"][:n_peer] if skin else None, rad_lw=ffull["rad_lw"][:n_peer] if skin else None,
                              out={k: v[:n_peer] for k, v in o1.items() if (k != "T_s" or skin)}, want_T_s=skin, check=False)
                    for _ in range(5):
                        s1.compute(1, zt, zu, *[ffull[k][:n_peer] for k in IN6], **kw)
                    torch.cuda.synchronize(root_dev)
                    t0 = time.perf_counter()
                    for _ in range(2 * a.steps):
                        s1.compute(1, zt, zu, *[ffull[k][:n_peer] for k in IN6], **kw)
                    torch.cuda.synchronize(root_dev)
                    slab_ms += (time.perf_counter() - t0) / (2 * a.steps) * 1e3
            peer_slab = {"rows": rows_peer_, "ms_per_step": round(slab_ms, 4), "ns_per_cell": round(slab_ms * 1e6 / n_peer, 5)}
        if not a.no_host_path and npass == 1:
            algo, skin = passes[0]
            with ab.Session(algo, ni, nj, 1, skin, precision=precision, device=devs, rows=rows) as sh:
                sh.set_humidity("sh")
                host_path = host_path_leg(ab, sh, ffull, skin, zt, zu, niter, ni * nj, {"f64": "float64"}.get(precision, "float32"), sync, lambda x: x)
            host_path["Mcell_s"] = round(ni * nj / host_path["ms_per_record"] / 1e3, 1)
            host_path["layout"] = f"one sharded session, {nsh} row blocks on devices {devs}: each device stages its rows over its own PCIe link"
        del ffull, o1
    except Exception as e:      # reports, never a reason to lose the line
        host_path = {"failed": str(e)}

    verify_msg = None
    if a.verify:
        step(True, nit=niter + 1)                              # other fields than every step before: a stale buffer cannot pass
        sync()
        b = (nstep[0] - 1) % nbuf
        ff = ab.synth_fields_device(ni, nj, precision=precision, device=root_dev, with_rad=True)
        bad = []
        for p, (algo, skin) in enumerate(passes):
            with ab.Session(algo, ni, nj, 1, skin, precision=precision, device=devs[root]) as s1:
                s1.set_humidity("sh")
                one = s1.compute(1, zt, zu, *[ff[k] for k in IN6], Niter=niter + 1, rad_sw=ff["rad_sw"] if skin else None,
                                 rad_lw=ff["rad_lw"] if skin else None, want_T_s=skin)
            bad += [f"{algo}:{k}" for k in gnames if k in one and not torch.equal(dst[b][p][k], one[k])]
        verify_msg = "gathered == single-GPU (bit-identical)" if not bad else f"MISMATCH in {bad}"
        if bad:
            raise SystemExit("verify failed: " + verify_msg)

    cells = ni * nj
    head_algo, head_skin = passes[0]
    bpc = algorithmic_bytes_per_cell(head_skin, esz)
    slow = max(range(len(shards)), key=lambda r: kms[r])
    n_slow = ni * shards[slow][1]
    achieved = bpc * n_slow / (kms[slow] * 1e-3) / 1e9 if kms[slow] > 0 else 0.0
    what = " + ".join(f"{al}{' + cool-skin/warm-layer' if sk else ''}" for al, sk in passes)
    headline = (a.config == 3 and head_algo == "coare3p6" and head_skin and (ni, nj) == (4320, 3600))
    ndist = len(set(devs))
    forced = os.environ.get("AEROBULK_AMD_GATHER") == "rccl"
    res = {
        "metric": "Mcell/s COARE3p6+cool-skin on 4320x3600 grid" if headline else f"Mcell/s {what} on {ni}x{nj} grid",
        "value": round(npass * cells * a.steps / elapsed / 1e6, 2), "unit": "Mcell/s", "n_gpus": nsh, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": dtype_label, "data": "synthetic", "launcher": "inprocess",
        "config": {"workload": f"BASELINE config {a.config}: {what}, {ni}x{nj} grid, nb_iter={niter}, zt=2 zu=10, one time record (jt=1=Nt)"
                               f"{' per algorithm, the ' + str(npass) + ' passes back-to-back' if npass > 1 else ''}, fields resident in the HBM of the "
                               f"device that owns their rows",
                   "grid": [ni, nj], "algo": head_algo if npass == 1 else list(ALL_ALGOS), "skin": head_skin if npass == 1 else any_skin, "nb_iter": niter,
                   "sharding": f"one process, one sharded session: {nsh} j-blocks of {shards[0][1]}..{shards[-1][1]} rows on devices {devs} "
                               f"(ab_session_compute_shards), {', '.join(gnames)} gathered to device {devs[root]} by ab_session_gather "
                               f"(in-process RCCL send / recv over {ndist if ndist > 1 else (1 if forced else 0)} communicator ranks; device-to-device "
                               f"copies for shards on the root's device), {nbuf} buffer set{'s' if nbuf > 1 else ''}"},
        "n_ranks_rccl": ndist if ndist > 1 else (1 if forced else 0),
        "resident": {"value": round(npass * cells * a.steps / elapsed_resident / 1e6, 2), "unit": "Mcell/s",
                     "ms_per_step": round(elapsed_resident / a.steps * 1e3, 4),
                     "note": "the same K steps with the fluxes left on the device that computed them (no gather)"},
        "per_device_kernel_ms": {f"shard{r}@gpu{shards[r][2]}": round(kms[r], 4) for r in range(len(shards))},
        "rows_per_shard": [sh[1] for sh in shards],
        "split": ("equal j-blocks" if rows is None else f"root-heavy: shard 0 (on the gather's destination) owns {rows[0]} rows, its {nsh - 1} peers {rows[1]} each"
                  + (" (--peer-rows)" if a.peer_rows > 0 else " (measured during set-up: ab_session_create_sharded_rows)")),
        "roofline": {"bound": "hbm", "limiter": "valu_fp64", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": None, "kernel": kernel_label(precision, head_algo, head_skin, n_slow, ab),
                     "kernel_ms": round(kms[slow], 4), "bytes_per_cell": bpc, "cells_per_launch": n_slow,
                     "note": "the slowest shard's launch of the first pass; per device, not summed over devices"},
    }
    res["limiter"] = LIMITER_NOTE
    res["calib"], res["value_norm"] = calib_record(calib_before, calib_after, res["value"])
    if host_path:
        res["host_path"] = host_path
    if one_gpu_ms:
        res["one_gpu"] = {"ms_per_step": round(one_gpu_ms, 4), "value": round(npass * cells / one_gpu_ms / 1e3, 2), "unit": "Mcell/s",
                          "note": f"the same workload as ONE unsharded session on device {devs[root]}, measured in this run: the numerator of the speed-ups"}
        res["speedup_vs_one_gpu"] = {"gathered": round(one_gpu_ms / (elapsed / a.steps * 1e3), 3), "resident": round(one_gpu_ms / (elapsed_resident / a.steps * 1e3), 3)}
        if nsh > 1 and peer_slab:
            t_cell = peer_slab["ns_per_cell"] * 1e-9          # a PEER's kernels (all passes) per cell, its slab alone on a device
            meas = [tune["link_GBps_per_peer"]] if (tune and ndist > 1) else []
            res.setdefault("gather", {})["model"] = gather_model(nsh, cells, t_cell, one_gpu_ms, npass * len(gnames) * esz, [60.0, 100.0, 130.0] + meas)
            res["gather"]["model"]["inputs"]["peer_slab"] = peer_slab
    if verify_msg:
        res["verify"] = verify_msg
    if tune:
        res["split_tuning"] = tune
    print(json.dumps(res), flush=True)
    for s_ in sessions:
        s_.close()



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5], help="BASELINE.json configuration (see the module docstring)")
    ap.add_argument("--algo", default="coare3p6")
    ap.add_argument("--no-skin", action="store_true")
    ap.add_argument("--skin", action="store_true", help="config 4: switch the skin scheme on for the three algorithms that have one")
    ap.add_argument("--niter", type=int, default=None)
    ap.add_argument("--no-nb-iter-8", action="store_true", help="skip the headline's companion pass at nb_iter = 8 (profiling runs: one kernel configuration per trace)")
    ap.add_argument("--grid", default=None)
    ap.add_argument("--precision", default=None, choices=["f64", "f32", "f32_storage", "f32_mixed"],
                    help="f32_storage: fp32 arrays with fp64 arithmetic (AB_F32_STORAGE); f32_mixed: fp32 arrays, fp64 anchors (SST, theta, "
                         "T_s, q, q_s, their differences, q_sat), fp32 transcendentals (AB_F32_MIXED: config 5's mode, inside the restated 1e-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlapped", action="store_true", help="skip the `overlapped` leg (independent consecutive records alternating on two streams)")
    ap.add_argument("--host-path", action="store_true", help="time the host-array leg for other configurations than the headline too")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-array leg (`host_path`: the reference's calling convention, pageable caller "
                                                                "arrays through AB_MEM_HOST, outside the timed region)")
    ap.add_argument("--resident", "--no-gather", dest="resident", action="store_true",
                    help="N>1: the fluxes stay on the GPU that computed them (what a GPU-resident ocean model consumes): no gather "
                         "in the timed region.  Without this flag the gathered run is the headline and the resident rate is "
                         "measured after it and reported as `resident`")
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
    ap.add_argument("--no-early-gather", action="store_true", help="N>1: rank 0 joins the gather of a chunk after computing its "
                                                                    "own chunk (instead of before)")
    ap.add_argument("--no-pipeline-gather", action="store_true", help="N>1: every step waits for its own gathers (round 2's order) instead "
                    "of letting them drain while the next step computes into the other buffer set")
    ap.add_argument("--chunks", type=int, default=4, help="N>1: row sub-blocks per rank (gather of one overlaps compute of the next)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "inprocess", "torchrun"],
                    help="N>1 started as plain `python bench.py --gpus N` (no WORLD_SIZE in the environment): inprocess (auto) = ONE process, one "
                         "sharded library session over the devices (ab_session_compute_shards + ab_session_gather: in-process RCCL), what a "
                         "Fortran host gets at mod_aerobulk.f90:250-262; torchrun = start `python -m torch.distributed.run` as a child (before "
                         "anything touches the GPU) and run one rank per GPU")
    ap.add_argument("--devices", default=None, help="inprocess: device ordinals of the shards, e.g. 0,1,2,3 (default 0..N-1); an ordinal may "
                                                     "repeat (several shards on one GPU: how the path is exercised on a one-GPU box)")
    a = ap.parse_args()

    if (a.gpus > 1 or a.launcher == "inprocess") and "WORLD_SIZE" not in os.environ:   # (--launcher inprocess --gpus 1: the same code path on one shard)
        if a.launcher == "torchrun":          # a child process, started before this one initialises the GPU; its exit code is ours
            import subprocess
            argv, skip = [], False
            for x in sys.argv[1:]:
                if skip or x.startswith("--launcher="):
                    skip = False
                elif x == "--launcher":
                    skip = True
                else:
                    argv.append(x)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", os.environ.get("MASTER_PORT", "29577"), os.path.abspath(__file__), *argv]
            raise SystemExit(subprocess.call(cmd))
        return main_inprocess(a)

    import torch
    import aerobulk_amd as ab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ngpu = torch.cuda.device_count()
    dev_index = local_rank if a.backend == "nccl" else local_rank % max(ngpu, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if a.backend == "nccl" else torch.device("cpu")     # where collective payloads live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    passes, grid, precision, niter = resolve_config(a)
    ni, nj = (int(x) for x in grid.lower().split("x"))
    npass = len(passes)
    any_skin = any(sk for _, sk in passes)
    zt, zu = 2.0, 10.0
    esz = 8 if precision == "f64" else 4
    tdt = torch.float64 if precision == "f64" else torch.float32
    dtype_label = {"f64": "f64", "f32": "f32", "f32_storage": "f32 arrays / f64 arithmetic",
                   "f32_mixed": "f32 arrays / f64 anchors + f32 transcendentals (AB_F32_MIXED)"}[precision]
    nout = 6 if any_skin else 5
    names = ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")[:nout]
    ngat = nout if a.gather_ts else 5            # fields that travel to rank 0
    gathered = world > 1 and not a.resident
    head_algo, head_skin = passes[0]

    # ---- sharding.  Without a gather: equal j-blocks.  With it: rank 0 (the destination) owns more rows than its peers
    # (shard_rows_root_heavy); how many is measured here unless --peer-rows says otherwise.
    tune = None
    if gathered:
        rows_peer = a.peer_rows if a.peer_rows > 0 else -(-nj // world)
        if a.peer_rows == 0:
            # (1) what one link delivers when all peers send at once: a few gathers of 32 MB per rank
            nb = 4 * 1024 * 1024
            tb = torch.zeros(nb, dtype=torch.float64, device=cdev)
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
            decision = torch.zeros(1, dtype=torch.int64, device=cdev)
            if rank == 0:
                try:
                    # (2) what one GPU computes: the same kernels on ~1 M cells of the same fields
                    rs = max(1, min(nj, (1 << 20) // ni))
                    fs = ab.synth_fields_device(ni, nj, 0, rs, precision=precision, device=dev, with_rad=True)
                    t_cell = 0.0
                    for algo, skin in passes:
                        with ab.Session(algo, ni, rs, 1, skin, precision=precision, device=dev_index) as ss:
                            ss.set_humidity("sh")
                            best = 1e9
                            for it in range(40):
                                ss.compute(1, zt, zu, *[fs[k] for k in IN6], Niter=niter, rad_sw=fs["rad_sw"] if skin else None,
                                           rad_lw=fs["rad_lw"] if skin else None, want_T_s=skin, check=False)
                                if it >= 30:
                                    best = min(best, ss.last_kernel_ms())
                        t_cell += best * 1e-3 / (ni * rs)
                    link = nb * 8 / t_gather
                    rows_peer = balanced_peer_rows(nj, world, t_cell, npass * ngat * esz, link)
                    tune = {"link_GBps": round(link / 1e9, 1), "kernel_Mcell_per_s": round(npass * 1e-6 / t_cell, 1)}
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

    # synthetic inputs generated straight into HBM; outputs packed [chunk, pass, field, cell]: ONE gather per
    # chunk and pass
    f = ab.synth_fields_device(ni, nj, j0, max(njl, 1), precision=precision, device=dev, with_rad=True)
    # Two sets of output buffers (and of rank 0's receive buffers): step t computes into set t % 2 while the gathers of step t - 1 may
    # still be draining set (t - 1) % 2 — in a time loop a step then costs max(compute, ingress of rank 0), not their sum; a set is
    # only written again after its gathers of two steps ago were waited for (--no-pipeline-gather: wait at the end of every step)
    nbuf = 2 if gathered else 1
    outbuf = torch.zeros((nbuf, chunks, npass, nout, n_cpad), dtype=tdt, device=dev)
    gather_lists = None
    send0 = None
    if gathered and rank == 0:     # rank 0's rows stay where they are; it joins the collective with an empty payload
        gather_lists = [[[[torch.empty((ngat, n_gpad), dtype=tdt, device=cdev) for _ in range(world)] for _ in range(npass)]
                         for _ in range(chunks)] for _ in range(nbuf)]
        send0 = torch.zeros((ngat, n_gpad), dtype=tdt, device=cdev)

    work = []  # per chunk: None (no rows) or the list over passes of (session, inputs, rad, out, skin)
    for c in range(chunks):
        r0 = min(c * cr, njl)
        rows = max(min(cr, njl - r0), 0)
        if rows == 0:
            work.append(None)
            continue
        lo, hi = r0 * ni, (r0 + rows) * ni
        ins = [f[k][lo:hi] for k in IN6]
        per_pass = []
        for p, (algo, skin) in enumerate(passes):
            sess = ab.Session(algo, ni, rows, 1, skin, precision=precision, device=dev_index)
            sess.set_humidity("sh")
            rad = (f["rad_sw"][lo:hi], f["rad_lw"][lo:hi]) if skin else (None, None)
            out = [{k: outbuf[b, c, p, i, :rows * ni] for i, k in enumerate(names) if (k != "T_s" or skin)} for b in range(nbuf)]
            per_pass.append((sess, ins, rad, out, skin))
        work.append(per_pass)

    # Rank 0 contributes nothing to the gather (its rows stay where they are) but is the destination of every peer, and it owns
    # the largest block.  It therefore joins the gather of (chunk c, pass p) BEFORE computing its own (c, p) — the collective
    # then only waits for the peers: a peer's chunk lands while rank 0 is still computing, and the last gather of a step
    # overlaps rank 0's last kernel instead of following it.  The order of operations is the same for RCCL and for the gloo
    # stand-in (which moves host copies of the payloads): post / compute / wait.
    early = gathered and rank == 0 and not a.no_early_gather

    def compute(c, p, b=0, nit=None):
        w = work[c]
        if w is not None:
            sess, ins, rad, out, skin = w[p]
            sess.compute(1, zt, zu, *ins, Niter=niter if nit is None else nit, rad_sw=rad[0], rad_lw=rad[1], out=out[b], want_T_s=skin, check=False)

    inflight = [[] for _ in range(nbuf)]     # per buffer set: the gathers still reading / filling it
    nstep = [0]                              # steps taken so far: the set of the next step is nstep % nbuf

    def drain(b=None):
        for bb in (range(nbuf) if b is None else (b,)):
            for w_ in inflight[bb]:
                w_.wait()      # RCCL: stream-level wait (does not block the host); gloo: host wait
            inflight[bb] = []

    def step(with_gather=True, pipelined=True):
        b = nstep[0] % nbuf
        # --verify: consecutive steps compute DIFFERENT fields (one more pass of the iteration on odd steps), so that a gather that
        # read or filled the wrong buffer set while two were in flight cannot pass the bit-identity check
        nit = niter + (nstep[0] % 2) if a.verify else niter
        nstep[0] += 1
        drain(b)                              # the gathers of two steps ago (this set's): done before the set is written again
        pending = inflight[b]
        for c in range(chunks):
            for p in range(npass):
                g = gathered and with_gather
                if g and early:
                    pending.append(dist.gather(send0, gather_lists[b][c][p], dst=0, async_op=True))
                compute(c, p, b, nit)
                if g and not early:
                    if rank == 0:
                        pending.append(dist.gather(send0, gather_lists[b][c][p], dst=0, async_op=True))
                    else:
                        payload = outbuf[b, c, p, :ngat]
                        if a.backend != "nccl":          # gloo moves host memory: the copy waits for the kernel
                            payload = payload.cpu()
                        pending.append(dist.gather(payload, None, dst=0, async_op=True))
        if not pipelined:
            drain(b)                          # round 2's order: every step ends with its own gathers

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    region_events = []                       # HIP events on the launch stream around the K launches of the latest timed() call

    def timed(nsteps, with_gather, pipelined=True):
        drain()
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()                           # (the session launches on torch's current stream: api.Session.compute)
        for _ in range(nsteps):
            step(with_gather, pipelined)
        e1.record()
        drain()                               # the last steps' gathers belong to the timed region
        sync()
        el = time.perf_counter() - t0
        region_events[:] = [e0, e1]
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def assemble(p):
        """rank 0: global fields [ngat, ni*nj] of pass p: its own block (never gathered) + the peers' gathered chunk buffers."""
        glob = torch.empty((ngat, ni * nj), dtype=tdt, device=dev)
        b = (nstep[0] - 1) % nbuf                                 # the set of the last step taken
        for c in range(chunks):                                   # own rows
            r0 = min(c * cr, njl)
            rows = max(min(cr, njl - r0), 0)
            if rows:
                glob[:, r0 * ni:(r0 + rows) * ni] = outbuf[b, c, p, :ngat, :rows * ni]
        for r in range(1, world):
            rj0, rnjl, _ = shard_rows_root_heavy(nj, world, r, rows_peer)
            for c in range(chunks):
                r0 = min(c * cr_peer, rnjl)
                rows = max(min(cr_peer, rnjl - r0), 0)
                if rows:
                    glob[:, (rj0 + r0) * ni:(rj0 + r0 + rows) * ni] = gather_lists[b][c][p][r][:, :rows * ni].to(dev)
        return glob

    # pre-roll, part of the setup: the GPU raises its clocks during the first ~100 ms of sustained work (the first
    # configuration measured after start-up runs 5-8 % slow otherwise, profiles/r1_notes.md); then the W warm-up steps
    calib_before = calibrate_box(ab, dev_index, 20) if rank == 0 else None     # the box, before anything of the flux path runs
    for _ in range(max(4, 40 // npass) if ni * nj <= 4320 * 3600 else 4):          # the same count on every rank (step() contains the gather)
        step()
    sync()
    for _ in range(a.warmup):
        step()
    sync()
    # timed region: EXACTLY K steps, no host sync inside
    elapsed = timed(a.steps, True, not a.no_pipeline_gather)
    # average launch duration over the timed region itself: HIP events on the launch stream before the first and behind the last of the K
    # steps' launches, / (K x launches per step).  Includes the gaps between back-to-back launches; never exceeds the step's wall time.
    region_ms = region_events[0].elapsed_time(region_events[1]) / a.steps
    calib_after = calibrate_box(ab, dev_index, 5) if rank == 0 else None
    verify_globs, verify_niter = None, niter
    if a.verify and gathered:                 # what the last step gathered, before the other timed runs reuse the buffers
        verify_niter = niter + ((nstep[0] - 1) % 2)
        if rank == 0:
            verify_globs = [assemble(p) for p in range(npass)]
    for w in work:
        if w is not None:
            for t in w:
                t[0].check()
    # further numbers of a gathered run: the same K steps with every step ending in its own gathers (round 2's order: compute and
    # ingress exposed one after the other), and with the fluxes left where they were computed
    elapsed_exposed = timed(a.steps, True, False) if (gathered and not a.no_pipeline_gather) else None
    elapsed_resident = timed(a.steps, False) if gathered else None
    # ... and the same without the row chunks either: a model that leaves its fluxes distributed launches ONE kernel per rank and
    # algorithm over the rank's whole block (the chunks exist to overlap the gather; on their own they only make the launches smaller)
    elapsed_resident_whole = None
    if gathered and chunks > 1 and njl > 0:
        whole = []
        obuf = torch.zeros((npass, nout, n_local), dtype=tdt, device=dev)
        for p, (algo, skin) in enumerate(passes):
            sess = ab.Session(algo, ni, njl, 1, skin, precision=precision, device=dev_index)
            sess.set_humidity("sh")
            whole.append((sess, [f[k][:n_local] for k in IN6], (f["rad_sw"][:n_local], f["rad_lw"][:n_local]) if skin else (None, None),
                          {k: obuf[p, i] for i, k in enumerate(names) if (k != "T_s" or skin)}, skin))

        def step_whole():
            for sess, ins, rad, out, skin in whole:
                sess.compute(1, zt, zu, *ins, Niter=niter, rad_sw=rad[0], rad_lw=rad[1], out=out, want_T_s=skin, check=False)
        for _ in range(3):
            step_whole()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step_whole()
        sync()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_resident_whole = float(t.item())
        for w_ in whole:
            w_[0].close()
    elif gathered and chunks > 1:      # (a rank without rows still joins the reduction)
        sync()
        t = torch.zeros(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_resident_whole = float(t.item())

    # the same whole-block launches with consecutive records alternating on two streams (N = 1: the headline's workload; N > 1: every rank its block)
    def ov_max(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x
    overlapped_s = None
    if not a.no_overlapped and (world == 1 or gathered):
        try:
            overlapped_s = overlapped_leg(ab, torch, passes, ni, njl, precision, dev_index, dev, f, names, tdt, niter, zt, zu, max(a.steps, 20) * 2, sync, ov_max)
        except Exception as e:      # a report, never a reason to lose the line
            overlapped_s = None
            if rank == 0:
                print(f"bench.py: overlapped leg failed: {e}", file=sys.stderr)

    # per-launch kernel duration: HIP events recorded by the library around each launch, on the launch stream.
    # Reading an event pair synchronises, so this is a separate pass over the same inputs (not in `elapsed`).
    kms = [0.0] * npass
    nrep = min(a.steps, 10)
    for _ in range(nrep):
        for w in work:
            if w is not None:
                for p, (sess, ins, rad, out, skin) in enumerate(w):
                    sess.compute(1, zt, zu, *ins, Niter=niter, rad_sw=rad[0], rad_lw=rad[1], out=out[0], want_T_s=skin, check=False)
                    kms[p] += sess.last_kernel_ms() / nrep
    # the roofline's kernel duration: one launch per step (N = 1, one pass, one chunk) -> the timed region's own average; otherwise the
    # event pair around a single launch of the first pass
    one_launch_per_step = (npass == 1 and chunks == 1)
    k_ms = region_ms if one_launch_per_step else kms[0]

    # the host-array leg: every rank moves ITS rows over its own PCIe link, all ranks at once (max over ranks per record)
    host_path = None
    if not a.no_host_path and npass == 1 and njl == 0 and (a.config == 3 or a.host_path):
        for _ in range(4):                    # a rank without rows still joins the other ranks' reductions (host_path_leg: records = 4)
            sync()
            t = torch.zeros(1, dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elif not a.no_host_path and npass == 1 and (a.config == 3 or a.host_path):
        try:
            def hp_max(x):
                if world > 1:
                    t = torch.tensor([x], dtype=torch.float64, device=cdev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    return float(t.item())
                return x
            with ab.Session(head_algo, ni, njl, 1, head_skin, precision=precision, device=dev_index) as sh:
                sh.set_humidity("sh")
                host_path = host_path_leg(ab, sh, {k: v[:n_local] for k, v in f.items()}, head_skin, zt, zu, niter, n_local,
                                          {"f64": "float64"}.get(precision, "float32"), sync, hp_max)
            host_path["Mcell_s"] = round(ni * nj / host_path["ms_per_record"] / 1e3, 1)
            host_path["layout"] = (f"{world} ranks, each staging its {njl}-row block over its own PCIe link at the same time (max over ranks per record)"
                                   if world > 1 else "one session, one device")
        except Exception as e:
            host_path = {"failed": str(e)}

    # BASELINE config 3 is quoted at nb_iter = 5 (the reference's default, mod_const.f90:33); its sea-ice series driver and BASELINE config 2 use 8
    # (src/ice/test_aerobulk_cdnf_series.f90:72): the same K steps at 8, reported beside the headline (never as `value`)
    alt8 = None
    if world == 1 and a.config == 3 and a.niter is None and not gathered and not a.no_nb_iter_8 and (head_algo, head_skin, ni, nj) == ("coare3p6", True, 4320, 3600):
        def step8():
            for w in work:
                if w is not None:
                    for sess, ins, rad, out, skin in w:
                        sess.compute(1, zt, zu, *ins, Niter=8, rad_sw=rad[0], rad_lw=rad[1], out=out[0], want_T_s=skin, check=False)
        for _ in range(40):          # (a pre-roll of its own: the host-array leg before it left the device idle long enough for its clocks to fall —
            step8()                  # the first lines of round 6 showed 3.00 ms per step here beside a kernel of 2.93)
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step8()
        sync()
        el8 = time.perf_counter() - t0
        k8 = 0.0
        for _ in range(nrep):
            step8()
            k8 += work[0][0][0].last_kernel_ms() / nrep
        alt8 = {"nb_iter": 8, "value": round(npass * ni * nj * a.steps / el8 / 1e6, 2), "unit": "Mcell/s", "ms_per_step": round(el8 / a.steps * 1e3, 4),
                "kernel_ms": round(k8, 4), "note": "the same K steps with nb_iter = 8 (as BASELINE config 2 and the reference's sea-ice series driver use); `value` above is BASELINE's nb_iter = 5, the reference's default"}

    verify_msg = None
    if a.verify and gathered and rank == 0:
        ff = ab.synth_fields_device(ni, nj, precision=precision, device=dev, with_rad=True)
        bad = []
        for p, (algo, skin) in enumerate(passes):
            glob = verify_globs[p]
            with ab.Session(algo, ni, nj, 1, skin, precision=precision, device=dev_index) as s1:
                s1.set_humidity("sh")
                one = s1.compute(1, zt, zu, *[ff[k] for k in IN6], Niter=verify_niter, rad_sw=ff["rad_sw"] if skin else None,
                                 rad_lw=ff["rad_lw"] if skin else None, want_T_s=skin)
            bad += [f"{algo}:{k}" for i, k in enumerate(names[:ngat]) if k in one and not torch.equal(glob[i], one[k])]
        verify_msg = "gathered == single-GPU (bit-identical)" if not bad else f"MISMATCH in {bad}"
        if bad:
            raise SystemExit("verify failed: " + verify_msg)

    if rank == 0:
        cells = ni * nj
        ms_per_step = elapsed / a.steps * 1e3
        value = npass * cells * a.steps / elapsed / 1e6
        bpc = algorithmic_bytes_per_cell(head_skin, esz)
        achieved = bpc * n_local / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pmc = committed_pmc(head_algo, head_skin, ni, nj, niter, precision) if world == 1 else None
        fresh = bool(pmc) and not pmc.get("stale")
        headline = (a.config == 3 and head_algo == "coare3p6" and head_skin and (ni, nj) == (4320, 3600))
        what = " + ".join(f"{al}{' + cool-skin/warm-layer' if sk else ''}" for al, sk in passes)
        prof = None
        if fresh:
            fl = pmc["fp64_insts_per_launch"]
            t_prof = pmc["kernel_us_pmc_pass"] * 1e-6
            tf = 64.0 * (2.0 * fl["fma"] + fl["mul"] + fl["add"] + fl["trans"]) / t_prof / 1e12
            prof = {"source": pmc["source"], "source_hash": pmc["source_hash"], "kernel_ms_in_profile": round(t_prof * 1e3, 4),
                    "hbm_traffic_bytes": round(pmc["traffic_bytes_per_launch"]),
                    "valu_insts_per_cell": round(pmc["valu_insts_per_cell"]), "valu_issue_frac": round(pmc["valu_busy"], 3),
                    "fp64_tflops": round(tf, 2), "fp64_peak_tflops": FP64_VECTOR_PEAK_TF, "fp64_frac": round(tf / FP64_VECTOR_PEAK_TF, 3)}
        res = {
            "metric": "Mcell/s COARE3p6+cool-skin on 4320x3600 grid" if headline else f"Mcell/s {what} on {ni}x{nj} grid",
            "value": round(value, 2), "unit": "Mcell/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": f"BASELINE config {a.config}: {what}, {ni}x{nj} grid, nb_iter={niter}, zt=2 zu=10, one time record "
                                   f"(jt=1=Nt){' per algorithm, the ' + str(npass) + ' passes back-to-back' if npass > 1 else ''}, "
                                   f"inputs/outputs resident in HBM",
                       "grid": [ni, nj], "algo": head_algo if npass == 1 else list(ALL_ALGOS), "skin": head_skin if npass == 1 else any_skin,
                       "nb_iter": niter,
                       "sharding": f"j-block x{world}" + (", fluxes stay on the GPU that computed them (no gather)" if (world > 1 and not gathered) else "")
                                   + ("" if not gathered else
                                      f": rank 0 owns {nj - (world - 1) * rows_peer} rows, ranks 1..{world - 1} {rows_peer} rows each "
                                      f"(rank 0 is the gather's destination; its compute is balanced against the peers' compute + transfer)"
                                      f" + {'RCCL' if a.backend == 'nccl' else 'gloo (host-staged stand-in)'} gather of {', '.join(names[:ngat])} to rank 0, "
                                      f"{chunks} overlapped row chunks per rank, rank 0 joins each gather {'before' if early else 'after'} computing its own chunk"),
                       **({"sharding_tuning": tune} if tune else {})},
            # The kernel is bound by fp64 VALU issue (hundreds of transcendentals per cell), not by HBM and not by MFMA (no
            # contraction on this path).  achieved/peak/frac are the HBM figures BASELINE.json asks for (algorithmic bytes over the
            # live kernel duration); `profile` holds the hardware-counter view of the binding resource, quoted only when the
            # committed profile was taken with this very device code.
            "roofline": {"bound": "hbm", "limiter": "valu_fp64", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": prof["hbm_traffic_bytes"] if prof else None,
                         "kernel": kernel_label(precision, head_algo, head_skin, n_local, ab),
                         "kernel_ms": round(k_ms, 4),
                         "kernel_ms_how": ("HIP events on the launch stream around the K launches of the timed region, / K" if one_launch_per_step
                                           else "HIP events around one launch of the first pass (separate pass over the same inputs)"),
                         "kernel_ms_single_launch": round(kms[0], 4),
                         "bytes_per_cell": bpc, "cells_per_launch": n_local,
                         "fp64_frac": prof["fp64_frac"] if prof else None,
                         "valu_issue_frac": prof["valu_issue_frac"] if prof else None,
                         "profile": prof if prof else ({"stale": "committed profile was taken with other device code: not quoted"} if pmc else None)},
        }
        res["limiter"] = LIMITER_NOTE
        res["calib"], res["value_norm"] = calib_record(calib_before, calib_after, value)
        if overlapped_s:
            res["overlapped"] = {"value": round(npass * cells / overlapped_s / 1e6, 2), "unit": "Mcell/s", "ms_per_step": round(overlapped_s * 1e3, 4),
                                 "note": ("consecutive INDEPENDENT records (single-record sessions) alternating on two streams through two sessions"
                                          + (", every rank its whole block, fluxes left where they were computed (no gather)" if world > 1 else "")
                                          + ": the next record's workgroups take each CU as the previous record's leave it — throughput of independent records, "
                                            "not the wall time of one record (`value`); a time loop carrying the warm layer's state cannot overlap its records")}
        if host_path:
            res["host_path"] = host_path
        if alt8:
            res["nb_iter_8"] = alt8
        if npass > 1:
            res["per_algorithm"] = {f"{al}{'+skin' if sk else ''}": {"kernel_ms": round(kms[p], 4),
                                                                        "Mcell_per_s": round(n_local / kms[p] / 1e3, 1) if kms[p] > 0 else None}
                                    for p, (al, sk) in enumerate(passes)}
        if gathered and tune and "kernel_Mcell_per_s" in tune:
            # the model's inputs are this run's own measurements: rank 0's kernel rate on ~1 M cells (a slab of a peer's size class) and the
            # link rate of the tuning gathers; the one-GPU step is NOT measured in a torchrun launch (every rank holds its rows only):
            # extrapolated from the same kernel rate, and said so
            t_cell = npass * 1e-6 / tune["kernel_Mcell_per_s"]
            gm = gather_model(world, cells, t_cell, t_cell * cells * 1e3, npass * ngat * esz, [60.0, 100.0, 130.0, float(tune["link_GBps"])])
            gm["inputs"]["one_gpu_step_ms_is"] = "extrapolated from the kernel rate on ~1 M cells (the driver's N = 1 line holds the measured one)"
        else:
            gm = None
        if gathered:
            res["gather"] = {"pipelined": not a.no_pipeline_gather, **({"model": gm} if gm else {}),
                             "note": ("two sets of output / receive buffers: the gathers of step t drain while step t + 1 computes; a set is "
                                      "written again only after its gathers were waited for" if not a.no_pipeline_gather else
                                      "every step ends with a wait for its own gathers")}
        if elapsed_exposed is not None:
            res["gather"]["unpipelined"] = {"value": round(npass * cells * a.steps / elapsed_exposed / 1e6, 2), "unit": "Mcell/s",
                                            "ms_per_step": round(elapsed_exposed / a.steps * 1e3, 4),
                                            "note": "the same K steps, every step waiting for its own gathers (round 2's order)"}
        if elapsed_resident is not None:
            res["resident"] = {"value": round(npass * cells * a.steps / elapsed_resident / 1e6, 2), "unit": "Mcell/s",
                               "ms_per_step": round(elapsed_resident / a.steps * 1e3, 4),
                               "note": "the same K steps with the fluxes left on the GPU that computed them (no gather)"}
            if elapsed_resident_whole:
                res["resident"]["chunked"] = dict(res["resident"])
                res["resident"].update(value=round(npass * cells * a.steps / elapsed_resident_whole / 1e6, 2),
                                       ms_per_step=round(elapsed_resident_whole / a.steps * 1e3, 4),
                                       note="K steps of ONE launch per rank and algorithm over the rank's whole block, fluxes left on the GPU that computed "
                                            "them: what a model with distributed fields gets (`chunked`: the same with the row chunks of the gathered run)")
        if verify_msg:
            res["verify"] = verify_msg
        res["precision_mode"] = {"f64": "AB_F64", "f32": "AB_F32", "f32_storage": "AB_F32_STORAGE", "f32_mixed": "AB_F32_MIXED"}[precision]
        if a.config == 5 and world == 1 and a.precision is None:
            # BASELINE config 5 says "fp32 path ... tolerance re-stated".  `value` is the AB_F32_MIXED session (fp32 arrays, fp64 anchors: inside
            # the restated 1e-4, tests/test_gpu_mixed.py) since round 3; rounds 1-2 timed AB_F32 (fp32 throughout: p99 2e-4, NOT inside it).
            # Both are measured here on the same fields so that the lines of different rounds can be compared.
            try:
                with ab.Session(head_algo, ni, njl, 1, head_skin, precision="f32", device=dev_index) as s32:
                    s32.set_humidity("sh")
                    o32 = {k: torch.empty(n_local, dtype=torch.float32, device=dev) for k in names}
                    args32 = dict(Niter=niter, rad_sw=f["rad_sw"] if head_skin else None, rad_lw=f["rad_lw"] if head_skin else None, out=o32,
                                  want_T_s=head_skin, check=False)
                    for _ in range(3):
                        s32.compute(1, zt, zu, *[f[k] for k in IN6], **args32)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        s32.compute(1, zt, zu, *[f[k] for k in IN6], **args32)
                    torch.cuda.synchronize()
                    el32 = time.perf_counter() - t0
                res["precision_modes"] = {
                    "AB_F32_MIXED": {"value": res["value"], "unit": "Mcell/s", "within_restated_tolerance": True,
                                     "note": "this line's `value`: BASELINE config 5 is compared against THIS mode since round 3"},
                    "AB_F32": {"value": round(cells * a.steps / el32 / 1e6, 2), "unit": "Mcell/s", "within_restated_tolerance": False,
                               "note": "fp32 arithmetic throughout: what BENCH_r01 / r02 config-5 lines timed; errors up to 1e-3 where theta - T_s or q - q_s is small"}}
            except Exception as e:
                res["precision_modes"] = {"failed": str(e)}
        if not a.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline_config1(niter, zt, zu) if a.config == 1 else cpu_baseline(head_algo, head_skin, niter, zt, zu)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "Mcell/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res), flush=True)
    for w in work:
        if w is not None:
            for t in w:
                t[0].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
