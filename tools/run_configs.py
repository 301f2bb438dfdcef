#!/usr/bin/env python
"""Run bench.py for every BASELINE.json configuration that fits one GPU and collect the JSON lines (SURVEY §8d "configs ->
concrete runs"): writes gpurun_out/configs.jsonl (copy to profiles/r4_configs.jsonl)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ("config 1: ncar, no skin, 360x180, nb_iter=5 (the reference's CPU case; reference timed on the same workload)", ["--config", "1", "--with-cpu-baseline"]),
    ("config 2: coare3p6, no skin, 1440x1080, nb_iter=8", ["--config", "2"]),
    ("config 3: coare3p6 + skin, 4320x3600, nb_iter=5 (headline)", []),
    ("config 3: coare3p6 + skin, 4320x3600, nb_iter=8", ["--niter", "8"]),
    ("config 4 (1 GPU): five algorithms back-to-back, no skin", ["--config", "4"]),
    ("config 4 (1 GPU): five algorithms back-to-back, skin where supported", ["--config", "4", "--skin"]),
    ("config 5 (1 GPU): ecmwf + skin, fp32 arrays in the mixed mode (AB_F32_MIXED: the timed path), 12960x10800", ["--config", "5", "--steps", "5"]),
    ("config 5 path on the 4320x3600 grid", ["--config", "5", "--grid", "4320x3600"]),
    ("config 5 (1 GPU) with fp32 arithmetic throughout (AB_F32: outside the restated tolerance)", ["--config", "5", "--precision", "f32", "--steps", "5"]),
    ("config 5 (1 GPU) with fp64 arithmetic on the fp32 arrays (AB_F32_STORAGE)", ["--config", "5", "--precision", "f32_storage", "--steps", "5"]),
]


def main():
    out = os.path.join(ROOT, "gpurun_out", "configs.jsonl")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        for label, extra in RUNS:
            base = [] if "--with-cpu-baseline" in extra else ["--no-cpu-baseline"]
            extra = [e for e in extra if e != "--with-cpu-baseline"]
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *base, *extra], capture_output=True, text=True)
            line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(label, "FAILED", pr.stderr[-500:])
                continue
            d = json.loads(line[-1])
            d["label"] = label
            fh.write(json.dumps(d) + "\n")
            print(f"{label}: {d['value']} {d['unit']}, {d['ms_per_step']} ms/step, kernel {d['roofline']['kernel_ms']} ms, "
                  f"{d['roofline']['achieved']} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
