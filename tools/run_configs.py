#!/usr/bin/env python
"""Run bench.py for every BASELINE.json configuration that fits one GPU and collect the JSON lines (SURVEY §8d "configs ->
concrete runs"): writes gpurun_out/configs.jsonl (copy to profiles/r1_configs.jsonl)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ("config 2: coare3p6, no skin, 1440x1080, nb_iter=8", ["--algo", "coare3p6", "--no-skin", "--grid", "1440x1080", "--niter", "8"]),
    ("config 3: coare3p6 + skin, 4320x3600, nb_iter=5 (headline)", []),
    ("config 3: coare3p6 + skin, 4320x3600, nb_iter=8", ["--niter", "8"]),
    ("config 4 (1 GPU): coare3p0 + skin", ["--algo", "coare3p0"]),
    ("config 4 (1 GPU): ecmwf + skin", ["--algo", "ecmwf"]),
    ("config 4 (1 GPU): coare3p0, no skin", ["--algo", "coare3p0", "--no-skin"]),
    ("config 4 (1 GPU): coare3p6, no skin", ["--algo", "coare3p6", "--no-skin"]),
    ("config 4 (1 GPU): ncar", ["--algo", "ncar"]),
    ("config 4 (1 GPU): ecmwf, no skin", ["--algo", "ecmwf", "--no-skin"]),
    ("config 4 (1 GPU): andreas", ["--algo", "andreas"]),
    ("config 5 (1 GPU): ecmwf + skin, fp32, 12960x10800", ["--algo", "ecmwf", "--precision", "f32", "--grid", "12960x10800", "--steps", "5"]),
]


def main():
    out = os.path.join(ROOT, "gpurun_out", "configs.jsonl")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        for label, extra in RUNS:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *extra], capture_output=True, text=True)
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
