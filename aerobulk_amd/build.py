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
HIPFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast-honor-pragmas",
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
    srcs = [os.path.join(CSRC, f) for f in ("ab_kernels.hip", "ab_turb_kernels.hip", "ab_ice_kernels.hip", "ab_phymbl.hip", "ab_calib.hip", "ab_runtime.hip", "ab_sharded.hip", "ab_cxx.cpp")]
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


FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
FFLAGS = ["-O2", "-fdefault-real-8"]     # default reals promoted like every build macro of the reference (arch/make.macro_GnuLinux:17)
FDIR = os.path.join(PKG, "fortran")
FLIB = os.path.join(FDIR, "libaerobulk_amd_fortran.a")
# module sources in dependency order: what the reference ships as lib/libaerobulk.a + mod/*.mod (its Makefile:22-42,66-71)
FMODS = ["mod_const", "mod_phymbl", "mod_aerobulk", "mod_blk_turb", "mod_blk_ice"]
# the repo's own drivers (test harnesses of tests/test_turb_series.py, test_neutral10.py, test_sea_ice.py, test_gpu_hosts.py,
# test_phymbl.py)
FDRIVERS = ["example_call_aerobulk", "turb_series_driver", "neutral10_driver", "turb_ice_driver", "phymbl_driver", "oce_ice_driver", "skin_driver"]
REF = os.environ.get("AEROBULK_REFERENCE", "/root/reference")
# Callers of the reference compiled UNCHANGED, where they lie, against the modules above (the drop-in check of SURVEY §8b): binaries
# go to oracle/_ref/dropin/ (git-ignored, travels to the GPU box like the other reference builds).  Build container only.
REF_CALLERS = ["src/tests/example_call_aerobulk.f90", "src/tests/test_cx_vs_wind.f90", "src/tests/test_coef_n10.f90",
               "src/tests/aerobulk_toy.F90", "src/tests/test_phymbl.f90", "src/ice/test_ice.f90", "src/ice/test_aerobulk_ice.f90",
               "src/ice/test_aerobulk_oce+ice.f90"]
DROPIN = os.path.join(ROOT, "oracle", "_ref", "dropin")


def _link_flags(exe_dir):
    rel = os.path.relpath(PKG, exe_dir)
    return ["-L", PKG, "-laerobulk_amd", f"-Wl,-rpath,$ORIGIN/{rel}"]


def build_fortran_host(force=False):
    """From-scratch Fortran host: mod_const, mod_phymbl, mod_aerobulk, mod_blk_* (TURB_*), the sea-ice modules, as
    libaerobulk_amd_fortran.a + *.mod, and the repo's drivers; needs amdflang."""
    if not os.path.exists(FC):
        print("amdflang absent: Fortran host not built")
        return None
    srcs = [os.path.join(FDIR, m + ".f90") for m in FMODS]
    objs = [os.path.join(FDIR, m + ".o") for m in FMODS]
    if force or _newer(FLIB, srcs + [os.path.abspath(__file__)]):
        for src, obj in zip(srcs, objs):      # serial: each module needs the .mod files of the previous ones
            _run([FC, *FFLAGS, "-fPIC", "-module-dir", FDIR, "-I", FDIR, "-c", src, "-o", obj])
        if os.path.exists(FLIB):
            os.remove(FLIB)
        _run(["/opt/rocm/lib/llvm/bin/llvm-ar", "rcs", FLIB, *objs])
    for d in FDRIVERS:
        src, exe = os.path.join(FDIR, d + ".f90"), os.path.join(FDIR, d + ".x")
        if force or _newer(exe, [src, FLIB, LIB]):
            _run([FC, *FFLAGS, "-I", FDIR, src, FLIB, *_link_flags(FDIR), "-o", exe])
    return FLIB


def build_reference_callers(force=False):
    """The reference's own drivers, unmodified, compiled from /root/reference against THIS repo's modules and library."""
    if not os.path.exists(FC) or not os.path.isdir(os.path.join(REF, "src")):
        print("reference tree or amdflang absent: reference callers not built (prebuilt oracle/_ref/dropin/*.x are used if present)")
        return []
    os.makedirs(DROPIN, exist_ok=True)
    built = []
    for rel in REF_CALLERS:
        src = os.path.join(REF, rel)
        exe = os.path.join(DROPIN, os.path.splitext(os.path.basename(rel))[0].replace("+", "_") + ".x")
        if force or _newer(exe, [src, FLIB, LIB]):
            try:
                _run([FC, *FFLAGS, "-I", FDIR, "-module-dir", DROPIN, src, FLIB, *_link_flags(DROPIN), "-o", exe])
            except subprocess.CalledProcessError:
                # drivers written in a dialect amdflang rejects whatever they are linked against (iargc() undeclared ...): reported,
                # not fatal; the API example must build
                if rel == REF_CALLERS[0]:
                    raise
                print(f"reference caller {rel}: does not compile with {os.path.basename(FC)} (see message above)", flush=True)
                continue
        built.append(exe)
    # the C++ example against include/aerobulk.hpp
    src = os.path.join(REF, "src", "tests", "example_call_aerobulk.cpp")
    exe = os.path.join(DROPIN, "example_call_aerobulk_cxx.x")
    if force or _newer(exe, [src, LIB, os.path.join(ROOT, "include", "aerobulk.hpp")]):
        _run(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"), src, *_link_flags(DROPIN), "-o", exe])
    built.append(exe)
    return built


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
    build_reference_callers(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
