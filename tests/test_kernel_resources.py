"""Build-time properties of the flux kernels that the performance rests on (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed):
no kernel spills to scratch, and the LDS of the blocks a CU is meant to hold (Tile::kOcc, ab_tile.hpp) fits its 160 KB — the fp64
kernels with the skin schemes sit 1 KB under that limit since the psi / e_sat tables went into LDS, and a kernel that slips over it
silently loses a quarter of its occupancy (measured: -3 %, profiles/r2_notes.md)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def remarks(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    from aerobulk_amd import build as b
    flags = getattr(b, "HIPFLAGS", None)
    if flags is None:
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast-honor-pragmas", "-fno-gpu-rdc", "-Wno-unused-function",
                 "-mllvm", "-disable-machine-licm", "-Xclang", "-target-feature", "-Xclang", "-fmacf64-inst"]
    out = tmp_path_factory.mktemp("res") / "k.o"
    pr = subprocess.run([HIPCC, *flags, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "aerobulk_amd", "csrc", "ab_kernels.hip"),
                         "-o", str(out), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-2000:]
    kernels, cur = {}, None
    for line in pr.stderr.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = kernels.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    flux = {k: v for k, v in kernels.items() if "flux_kernelI" in k}
    assert len(flux) == 64, len(flux)     # 16 (5 algorithms x skin where it exists, x DIAG) x {fp64, fp32, fp32 arrays / fp64 arithmetic, mixed}
    cu = {k: v for k, v in kernels.items() if "flux_kernel_cuI" in k}
    assert len(cu) == 4, len(cu)          # one workgroup per CU: COARE 3.0 / 3.6 with the skin schemes, fp64 arithmetic on fp64 / fp32 arrays
    flux.update(cu)
    return flux


def _params(name):
    """flux_kernel<R, ALGO, SKIN, DIAG, S, A> from the mangled name (A: anchor type; R = f with A = d is the mixed mode)"""
    m = re.search(r"flux_kernelI([df])Li(\d)ELb([01])ELb([01])E([df])([df])E", name)
    return m.group(1), int(m.group(2)), m.group(3) == "1", m.group(4) == "1", m.group(5), m.group(6)


def test_no_flux_kernel_uses_scratch(remarks):
    bad = {k: v["ScratchSize"] for k, v in remarks.items() if v["ScratchSize"]}
    assert not bad, bad


def test_lds_of_the_resident_blocks_fits_the_cu(remarks):
    seen = set()
    for name, v in remarks.items():
        if "flux_kernel_cuI" in name:
            # ONE workgroup of sixteen waves per CU: all of its 160 KB, four waves per SIMD (128 VGPRs)
            assert v["Occupancy"] >= 4 and 150 * 1024 < v["LDS"] <= LDS_PER_CU, (name, v)
            continue
        r, algo, skin, diag, s, a = _params(name)
        if algo == 3:
            assert v["LDS"] < (6144 if r == "d" else 4096)   # NCAR: direct kernel, math tables (+ fp64: the Kansas psi_m / psi_h pair, 3.5 KB)
            continue
        if r == "d":
            occ = 4 if (skin or diag or algo in (1, 2)) else 5  # Tile::kOcc; the DIAG instantiations are launched for four waves, COARE without skin too (its bit-indexed psi tables: 12.8 KB)
            if diag and skin:
                occ = 3                       # the diagnostics of the kernels with the skin schemes on top of lean kernels that fill their 128 registers
        elif a == "d":                        # mixed: fp32 work, fp64 anchors
            occ = 4 if diag else ((6 if algo == 4 else 5) if skin else 7)
        else:
            occ = 4 if diag else ((6 if algo == 4 else 7) if skin else 8)
        assert v["Occupancy"] >= occ, (name, v)                       # registers allow the designed occupancy ...
        assert v["LDS"] * occ <= LDS_PER_CU, (name, v, occ)           # ... and so does the LDS
        if r == "d" and not diag:
            # LDS is allocated in granules of 512 B (tools/micro/lds_granule.hip, profiles/r3_lds_granule.txt: 27 136 B -> six blocks per
            # CU, 27 307 B -> five; 32 768 B -> five, 32 769 B -> four): no hidden cliff with the rounded size
            assert (v["LDS"] + 511) // 512 * 512 * occ <= LDS_PER_CU, (name, v, occ)
        seen.add((r, a, skin, diag))
    assert len(seen) == 12
