"""Box calibration (ab_calibrate, aerobulk_amd/csrc/ab_calib.hip) and what bench.py derives from it: `calib`, `value_norm`, `limiter` — the keys that let
two benchmark lines taken on different leases be compared (README "Comparing rounds").  SURVEY §8d; no counterpart in the reference."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_calib_record_normalises_to_the_reference_clock():
    sys.path.insert(0, ROOT)
    import bench
    rec, norm = bench.calib_record({"fma_f64_tflops": 70.0, "sclk_mhz": 2100}, {"fma_f64_tflops": 72.0, "sclk_mhz": 2150}, 7100.0)
    assert rec["fma_f64_tflops_before"] == 70.0 and rec["fma_f64_tflops_after"] == 72.0 and rec["reference_tflops"] == bench.CALIB_REF_TFLOPS
    assert abs(norm - 7100.0 * bench.CALIB_REF_TFLOPS / 71.0) < 0.01 and abs(rec["box_speed"] - 71.0 / bench.CALIB_REF_TFLOPS) < 1e-4
    rec, norm = bench.calib_record({"fma_f64_tflops": None, "error": "x"}, None, 7100.0)        # a failed calibration never costs the line
    assert norm is None and rec["fma_f64_tflops_before"] is None


def test_gather_model_gives_the_numbers_design_md_quotes():
    """DESIGN.md §6's table of the gathered N = 8 run: the model bench.py prints (`gather.model`) from three stated inputs."""
    sys.path.insert(0, ROOT)
    import bench
    cells, t_cell, one_gpu_ms = 4320 * 3600, 0.2790e-3 / (4320 * 450), 1.9473          # this round's kernel: 4320x450 in 0.279 ms, full grid 1.947 ms
    m = bench.gather_model(8, cells, t_cell, one_gpu_ms, 40, [60.0, 100.0, 130.0])
    sp = {k: v["speedup_vs_one_gpu"] for k, v in m["predicted"].items()}
    assert abs(sp["60 GB/s per peer"] - 2.19) < 0.03 and abs(sp["100 GB/s per peer"] - 3.06) < 0.04 and abs(sp["130 GB/s per peer"] - 3.72) < 0.05, sp
    assert abs(m["gather_free_ceiling"] - 6.98) < 0.02
    assert abs(m["predicted"]["60 GB/s per peer"]["root_share"] - 0.399) < 0.003
    # fast links: the equal split, and the gather-free ceiling
    fast = bench.gather_model(8, cells, t_cell, one_gpu_ms, 40, [1e6])["predicted"]["1e+06 GB/s per peer"]
    assert abs(fast["peer_share"] - 0.125) < 1e-4 and abs(fast["speedup_vs_one_gpu"] - 6.98) < 0.02
    assert bench.balanced_peer_rows(3600, 8, t_cell, 40, 60e9) == round(0.0859 * 3600)


def test_committed_profiles_valu_instructions_per_cell_did_not_rise():
    """The headline kernel's VALU instructions per cell of this round's committed counter profile against the previous round's: a rise of more than
    0.5 % is a regression of the kernel, whatever the box's clocks did to the Mcell/s."""
    prof = os.path.join(ROOT, "profiles")
    cur, prev = os.path.join(prof, "r6_pmc.json"), os.path.join(prof, "r5_pmc.json")
    if not os.path.exists(cur):
        pytest.skip("profiles/r6_pmc.json not taken yet")
    a, b = json.load(open(cur)), json.load(open(prev))
    assert a["config"] == b["config"]
    assert a["valu_insts_per_cell"] <= 1.005 * b["valu_insts_per_cell"], (a["valu_insts_per_cell"], b["valu_insts_per_cell"])


@pytest.mark.gpu
def test_calibration_kernels_report_plausible_rates():
    import aerobulk_amd as ab
    for _ in range(20):                                                        # ~100 ms of work: the clocks are up (the first launches of a cold
        ab.calibrate("fma_f64")                                                # device gave 50 and 57 TFLOP/s)
    ms, tf = ab.calibrate("fma_f64")
    assert 0.5 < ms < 20.0 and 40.0 < tf < 95.0, (ms, tf)                       # 78.6 TFLOP/s is the guide's peak at 2.4 GHz; round-6 leases: 66
    ms2, tf2 = ab.calibrate("fma_f64")
    assert abs(tf2 - tf) < 0.03 * tf                                           # ... and repeatably
    ms, gbs = ab.calibrate("hbm_copy")
    assert 1500.0 < gbs < 8000.0, (ms, gbs)


@pytest.mark.gpu
def test_bench_line_carries_calib_value_norm_and_limiter():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "1440x1080", "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                          "--no-nb-iter-8"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    c = res["calib"]
    assert 40.0 < c["fma_f64_tflops_before"] < 95.0 and 40.0 < c["fma_f64_tflops_after"] < 95.0 and c["reference_tflops"] == 66.0
    assert abs(res["value_norm"] - res["value"] * 66.0 / (0.5 * (c["fma_f64_tflops_before"] + c["fma_f64_tflops_after"]))) < 0.02 * res["value"]
    assert res["limiter"].startswith("valu_fp64") and res["roofline"]["bound"] == "hbm"
    # independent records alternating on two streams: never slower than one record after the other (a 0.14 ms launch gains most)
    assert res["overlapped"]["value"] > 0.97 * res["value"] and "not the wall time of one record" in res["overlapped"]["note"]
