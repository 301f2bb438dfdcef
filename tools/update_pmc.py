#!/usr/bin/env python
"""Turn a tools/prof_quick.sh run (gpurun_out/prof_<tag>/) into the committed artefacts:
    profiles/<name>_flux_kernel.txt  (rocpd summaries of the 5 passes)   and   profiles/r4_pmc.json (what bench.py quotes,
stamped with the hash of the kernel sources + flags it was taken with: bench.py only quotes it for that very device code).

    python tools/update_pmc.py gpurun_out/prof_<tag> <name>
"""
import contextlib
import io
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary  # noqa: E402


def main():
    d, name = sys.argv[1], sys.argv[2]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for p in ("stats", "pmc_sq", "pmc_f64", "pmc_fetch", "pmc_write"):
            rocpd_summary.summarise(os.path.join(d, p, "bench_results.db"))
    txt = "\n".join(l for l in buf.getvalue().splitlines() if l.startswith("==") or l.startswith("kernel ") or "flux_kernel" in l or l.startswith("-- PMC"))
    out = os.path.join(ROOT, "profiles", f"{name}_flux_kernel.txt")
    open(out, "w").write(txt + "\n")
    val = lambda c: float(re.search(rf"flux_kernel\S+\s+{c}\s+([0-9.]+)", txt).group(1))
    durs = [float(m) for m in re.findall(r"flux_kernel\S+\s+\d+\s+([0-9.]+)", txt)]     # avg us per pass: stats, pmc_sq, pmc_f64, fetch, write
    avg_us = durs[0]
    sys.path.insert(0, ROOT)
    import bench
    cells = 4320 * 3600
    pmc = {
        "source": f"profiles/{name}_flux_kernel.txt (rocprofv3 --pmc, MI355X, COARE3p6+skin 4320x3600 fp64 nb_iter=5; tools/prof_quick.sh)",
        "config": {"algo": "coare3p6", "skin": True, "grid": [4320, 3600], "nb_iter": 5, "precision": "f64"},
        "traffic_bytes_per_launch": (2 * val("FETCH_SIZE") + val("WRITE_SIZE")) * 1024,   # gfx950: FETCH_SIZE counts 2x too little
        "valu_insts_per_cell": val("SQ_INSTS_VALU") * 64 / cells,   # wave instructions x 64 lanes / cells
        "valu_busy": val("SQ_ACTIVE_INST_VALU") * 4 / (1024 * val("GRBM_GUI_ACTIVE") / 8),
        "kernel_us_rocprof_avg": avg_us,
        "kernel_us_timed_region": float(re.search(r"last \d+ dispatches[^:]*: avg_us ([0-9.]+)", txt).group(1)),   # without the pre-roll / warm-up launches
        "kernel_us_pmc_pass": durs[2] if len(durs) > 2 else avg_us,   # duration in the pass that counted the fp64 instructions
        "source_hash": bench.kernel_source_hash(),
        "fp64_insts_per_launch": {k: val(f"SQ_INSTS_VALU_{k.upper()}_F64") for k in ("fma", "mul", "add", "trans")},
        "waves_per_launch": val("SQ_WAVES"),
    }
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "r4_pmc.json"), "w"), indent=1)
    print(out)
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
