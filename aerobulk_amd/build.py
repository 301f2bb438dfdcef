"""Build the in-tree native libraries (gfx950 only).

    python -m aerobulk_amd.build          # libaerobulk_amd.so (+ C++ wrapper, Fortran host if amdflang exists)

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libaerobulk_amd.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
# -disable-machine-licm: keep the s_mov of polynomial coefficients next to their use.  Hoisted out of the iteration loop
# they exhaust the 102 SGPRs and come back as v_readlane/v_mov_b64 VALU traffic (-13 % VALU instructions, profiles/r1_notes.md)
# -target-feature -fmacf64-inst: without the 2-address v_fmac_f64 the Horner steps a*b + C are selected as the 3-address
# v_fma_f64 with the coefficient C read straight from an SGPR pair; with it every step pays a v_mov_b64 (VALU) to bring C
# into the accumulator register (10 % of the VALU instructions of the iteration loop, profiles/r1_notes.md)
HIPFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
            "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-mllvm", "-disable-machine-licm",
            "-Xclang", "-target-feature", "-Xclang", "-fmacf64-inst"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def build_engine(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("ab_kernels.hip", "ab_turb_kernels.hip", "ab_ice_kernels.hip", "ab_runtime.hip", "ab_sharded.hip", "ab_cxx.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + [
        os.path.join(ROOT, "include", "aerobulk_amd.h"), os.path.join(ROOT, "include", "aerobulk.hpp"), os.path.abspath(__file__)]
    if not force and not _newer(LIB, deps):
        return LIB
    objs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        if force or _newer(o, deps):
            extra = ["-x", "hip"] if s.endswith(".cpp") else []
            _run([HIPCC, *HIPFLAGS, "-I", os.path.join(ROOT, "include"), *extra, "-c", s, "-o", o])
        objs.append(o)
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


def build_fortran_host(force=False):
    """From-scratch Fortran host module (mod_aerobulk) + example driver; needs amdflang."""
    fc = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
    if not os.path.exists(fc):
        print("amdflang absent: Fortran host not built")
        return None
    fdir = os.path.join(PKG, "fortran")
    src = os.path.join(fdir, "mod_aerobulk.f90")
    drv = os.path.join(fdir, "example_call_aerobulk.f90")
    exe = os.path.join(fdir, "example_call_aerobulk.x")
    if not os.path.exists(src):
        return None
    if force or _newer(exe, [src, drv, LIB]):
        _run([fc, "-O2", "-fdefault-real-8", "-module-dir", fdir, "-c", src, "-o", os.path.join(fdir, "mod_aerobulk.o")])
        _run([fc, "-O2", "-fdefault-real-8", "-I", fdir, drv, os.path.join(fdir, "mod_aerobulk.o"),
              "-L", PKG, "-laerobulk_amd", "-Wl,-rpath,$ORIGIN/..", "-o", exe])
    # TURB_* modules (mod_blk_coare3p6 ...) + the station time-series driver
    tsrc = os.path.join(fdir, "mod_blk_turb.f90")
    tdrv = os.path.join(fdir, "turb_series_driver.f90")
    texe = os.path.join(fdir, "turb_series_driver.x")
    if force or _newer(texe, [src, tsrc, tdrv, LIB]):
        _run([fc, "-O2", "-fdefault-real-8", "-module-dir", fdir, "-I", fdir, "-c", tsrc, "-o", os.path.join(fdir, "mod_blk_turb.o")])
        _run([fc, "-O2", "-fdefault-real-8", "-I", fdir, tdrv, os.path.join(fdir, "mod_blk_turb.o"),
              os.path.join(fdir, "mod_aerobulk.o"), "-L", PKG, "-laerobulk_amd", "-Wl,-rpath,$ORIGIN/..", "-o", texe])
    ndrv = os.path.join(fdir, "neutral10_driver.f90")
    nexe = os.path.join(fdir, "neutral10_driver.x")
    if force or _newer(nexe, [src, tsrc, ndrv, LIB]):
        _run([fc, "-O2", "-fdefault-real-8", "-I", fdir, ndrv, os.path.join(fdir, "mod_blk_turb.o"),
              os.path.join(fdir, "mod_aerobulk.o"), "-L", PKG, "-laerobulk_amd", "-Wl,-rpath,$ORIGIN/..", "-o", nexe])
    # sea-ice modules (mod_blk_ice_nemo ...) + their driver
    isrc = os.path.join(fdir, "mod_blk_ice.f90")
    idrv = os.path.join(fdir, "turb_ice_driver.f90")
    iexe = os.path.join(fdir, "turb_ice_driver.x")
    if force or _newer(iexe, [src, isrc, idrv, LIB]):
        _run([fc, "-O2", "-fdefault-real-8", "-module-dir", fdir, "-I", fdir, "-c", isrc, "-o", os.path.join(fdir, "mod_blk_ice.o")])
        _run([fc, "-O2", "-fdefault-real-8", "-I", fdir, idrv, os.path.join(fdir, "mod_blk_ice.o"),
              os.path.join(fdir, "mod_aerobulk.o"), "-L", PKG, "-laerobulk_amd", "-Wl,-rpath,$ORIGIN/..", "-o", iexe])
    return exe


def build_cxx_example(force=False):
    """C++ API example driver (aerobulk::model), linked against libaerobulk_amd.so only."""
    src = os.path.join(CSRC, "example_call_aerobulk.cpp")
    exe = os.path.join(CSRC, "example_call_aerobulk_cxx.x")
    if force or _newer(exe, [src, LIB, os.path.join(ROOT, "include", "aerobulk.hpp")]):
        _run(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"), src, "-L", PKG, "-laerobulk_amd",
              "-Wl,-rpath,$ORIGIN/..", "-o", exe])
    return exe


def build_oracle():
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "all"])


def build_all(force=False):
    build_engine(force)
    build_fortran_host(force)
    build_cxx_example(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
