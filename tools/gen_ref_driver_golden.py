#!/usr/bin/env python3
"""Golden text of tests/test_reference_drivers.py: what the reference's own drivers print — src/ice/test_ice.f90 (oracle/_ref/ref_test_ice.x)
and the interactive toy driver (src/tests/aerobulk_toy.F90, unmodified,
linked with the reference's own library: oracle/_ref/ref_aerobulk_toy.x) prints for the inputs of the reference's test_algos.sh, and for
the same case with the skin schemes (-S) and with relative humidity (-r).  Build container only.  Data: stdin and stdout of the runs."""
import json
import os
import subprocess
import tempfile
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    {"name": "test_ice", "exe": "test_ice", "args": [], "stdin": "-10.\n"},                       # air temperature [deg.C]; 17-digit output
    {"name": "test_phymbl", "exe": "test_phymbl", "args": [], "stdin": "10.\n15.\n8.\n"},      # height [m], T [deg.C], q [g/kg]: theta / pressure branch
    {"name": "test_algos.sh", "args": [], "stdin": "10.\n2.\n22.\n20.\n12.\n5\n"},              # zu zt SST t_zt q_zt[g/kg] wind: the README's table
    {"name": "skin schemes", "args": ["-S"], "stdin": "10.\n2.\n22.\n20.\n12.\n5\n600.\n350.\n"},    # + rad_sw, rad_lw
    {"name": "relative humidity, stable", "args": ["-r"], "stdin": "10.\n2.\n22.\n25.\n80.\n9\n"},
]


def main():
    out = []
    for c in CASES:
        c.setdefault("exe", "aerobulk_toy")
        exe = os.path.join(ROOT, "oracle", "_ref", f"ref_{c['exe']}.x")
        if not os.path.exists(exe):
            sys.exit("make -C oracle all first (needs /root/reference)")
        with tempfile.TemporaryDirectory() as tmp:      # test_ice.f90 writes a .dat file into its working directory
            pr = subprocess.run([exe, *c["args"]], input=c["stdin"], capture_output=True, text=True, timeout=120, cwd=tmp)
        assert pr.returncode == 0, pr.stderr
        out.append(dict(c, stdout=pr.stdout))
        print(c["name"], len(pr.stdout.splitlines()), "lines")
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ref_toy_outputs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
