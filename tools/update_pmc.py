#!/usr/bin/env python
"""Turn a tools/prof_quick.sh run (gpurun_out/prof_<tag>/summary.txt) into the committed artefacts:
    profiles/<name>_flux_kernel.txt   the rocpd summaries of the five passes (statistics, two counter groups, FETCH_SIZE, WRITE_SIZE)
    profiles/<name>.json              per flux kernel of the run: duration, VALU instructions per cell, VALU busy, fp64 instruction mix,
                                      HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, in KiB: the gfx950 corrections of MI355X_MICROARCH.md),
                                      stamped with the hash of the kernel sources + flags (bench.py quotes the headline's file only for
                                      that very device code)

    python tools/update_pmc.py gpurun_out/prof_<tag> <name> --cells 15552000 [--headline]
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("name")
    ap.add_argument("--cells", type=int, default=4320 * 3600)
    ap.add_argument("--what", default="COARE3p6+skin 4320x3600 fp64 nb_iter=5")
    ap.add_argument("--headline", action="store_true", help="also the flat record bench.py quotes (bench.PMC_JSON: profiles/r6_pmc.json)")
    a = ap.parse_args()
    txt = open(os.path.join(a.dir, "summary.txt")).read()
    out_txt = os.path.join(ROOT, "profiles", f"{a.name}_flux_kernel.txt")
    open(out_txt, "w").write(txt)
    sys.path.insert(0, ROOT)
    import bench
    kernels = sorted(set(re.findall(r"^(\S*flux_kernel\S*)\s+\d+\s+[0-9.]+", txt, flags=re.M)))
    recs = {}
    for k in kernels:
        ke = re.escape(k)
        val = lambda c: float(m.group(1)) if (m := re.search(rf"^{ke}\s+{c}\s+([0-9.]+)", txt, flags=re.M)) else None
        durs = [float(x) for x in re.findall(rf"^{ke}\s+\d+\s+([0-9.]+)", txt, flags=re.M)]      # avg us per pass: stats, pmc_sq, pmc_f64, fetch, write
        tl = re.search(rf"^{ke}\s+last \d+ dispatches[^:]*: avg_us ([0-9.]+)", txt, flags=re.M)
        gui = val("GRBM_GUI_ACTIVE")
        rec = {
            "kernel_us_rocprof_avg": durs[0] if durs else None,
            "kernel_us_timed_region": float(tl.group(1)) if tl else None,
            "kernel_us_pmc_pass": durs[2] if len(durs) > 2 else None,
            "valu_insts_per_cell": val("SQ_INSTS_VALU") * 64 / a.cells if val("SQ_INSTS_VALU") else None,
            "valu_busy": val("SQ_ACTIVE_INST_VALU") * 4 / (1024 * gui / 8) if gui and val("SQ_ACTIVE_INST_VALU") else None,
            "wait_inst_any_frac": val("SQ_WAIT_INST_ANY") / val("SQ_WAVE_CYCLES") if val("SQ_WAVE_CYCLES") and val("SQ_WAIT_INST_ANY") else None,
            "traffic_bytes_per_launch": (2 * val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024 if val("FETCH_SIZE") and val("WRITE_SIZE") else None,
            "fp64_insts_per_launch": {x: val(f"SQ_INSTS_VALU_{x.upper()}_F64") for x in ("fma", "mul", "add", "trans")},
            "trans_f32_insts_per_launch": val("SQ_INSTS_VALU_TRANS_F32"),
            "waves_per_launch": val("SQ_WAVES"),
        }
        if rec["valu_busy"] and rec["valu_busy"] > 1.:      # the formula is calibrated on fp64 kernels (four cycles per issued instruction); fp32 kernels
            rec["valu_busy_uncalibrated"] = rec.pop("valu_busy")   # with more than four waves per SIMD exceed it: reported, not interpreted
            rec["valu_busy"] = None
        recs[k] = rec
    doc = {"source": f"profiles/{a.name}_flux_kernel.txt (rocprofv3 --pmc in separate passes, MI355X; tools/prof_quick.sh): {a.what}",
           "cells_per_launch": a.cells, "source_hash": bench.kernel_source_hash(), "kernels": recs}
    json.dump(doc, open(os.path.join(ROOT, "profiles", f"{a.name}.json"), "w"), indent=1)
    if a.headline:
        k = next(k for k in kernels if "flux_kernel_cu" in k) if any("flux_kernel_cu" in k for k in kernels) else kernels[0]
        flat = dict(recs[k], source=doc["source"], source_hash=doc["source_hash"], kernel=k,
                    config={"algo": "coare3p6", "skin": True, "grid": [4320, 3600], "nb_iter": 5, "precision": "f64"})
        json.dump(flat, open(bench.PMC_JSON, "w"), indent=1)           # (this round's file: bench.py names it)
    print(out_txt)
    print(json.dumps(doc, indent=1)[:3000])


if __name__ == "__main__":
    main()
