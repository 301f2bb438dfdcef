#!/usr/bin/env python
"""Does the scalar instruction stream cost the VALU-bound flux kernel anything?  The kernel issues 3 400 SALU instructions per cell
next to its 4 787 VALU instructions, 2 020 of them `s_mov_b32` that materialise fp64 constants (VOP3 on gfx9 takes no literal: every
coefficient of a polynomial is two scalar moves into an SGPR pair).  SALU has its own issue port, but a wave issues in order: while
it issues scalar moves it offers the VALU port nothing.

The probe builds variants of the library in which the compiled assembly of ONE kernel is rewritten (same registers, same results):
  dup   every `s_mov_b32 sN, <literal|inline>` is followed by a dead copy into s100 (the scalar-move count doubles)
  dup2  two dead copies
  nop   a `s_nop 0` after every scalar move instead (same code size growth, no SALU issue)
and tools/ab_compare.py times them against the unmodified build on the same box:

    python tools/salu_probe.py build            # here (no GPU): build/var/libab_salu_{base,dup,dup2,nop}.so
    gpurun -- python tools/ab_compare.py --configs coare3p6:1:5 salu_base salu_dup salu_dup2 salu_nop
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_profile as ip   # noqa: E402


def rewrite(lines, i0, i1, mode):
    out, n = [], 0
    for ln in lines[i0:i1]:
        out.append(ln)
        s = ln.strip()
        if mode != "base" and re.match(r"s_mov_b32\s+s\d+,\s*(0x[0-9a-f]+|-?\d+(\.\d+)?)\s*(;.*)?$", s):
            src = s.split(",", 1)[1].split(";")[0].strip()
            n += 1
            if mode == "nop":
                out.append("\ts_nop 0")
            else:
                out.append(f"\ts_mov_b32 s100, {src}")
                if mode == "dup2":
                    out.append(f"\ts_mov_b32 s101, {src}")
    return out, n


def build(kernel):
    var = os.path.join(ROOT, "build", "var")
    os.makedirs(var, exist_ok=True)
    flags = ip.hipflags()
    src = os.path.join(ip.CSRC, "ab_kernels.hip")
    inc = ["-I", os.path.join(ROOT, "include")]
    s_in = os.path.join(var, "salu.s")
    subprocess.check_call(["hipcc", *flags, *inc, "--cuda-device-only", "-S", src, "-o", s_in])
    lines = open(s_in).read().split("\n")
    os.remove(s_in)
    i0, i1, mangled, pretty = ip.find_kernel(lines, kernel)
    k = re.escape(mangled)
    for mode in ("base", "dup", "dup2", "nop"):
        new, n = rewrite(lines, i0, i1, mode)
        text = "\n".join(lines[:i0] + new + lines[i1:])
        m = re.search(r"\.amdhsa_kernel " + k + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        desc = m.group(1)
        ns = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", desc).group(1))
        assert ns <= 100, f"kernel uses {ns} SGPRs: s100/s101 are taken"
        d2 = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", desc)
        text = text.replace(desc, d2)
        text = re.sub(r"(\.set " + k + r"\.numbered_sgpr, )\d+", r"\g<1>102", text)
        tag = f"salu_{mode}"
        s_out = os.path.join(var, f"{tag}.s")
        open(s_out, "w").write(text)
        obj, hsaco, fb = (os.path.join(var, f"{tag}.{e}") for e in ("dev.o", "hsaco", "hipfb"))
        LLVM = ip.LLVM
        subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_out, "-o", obj])
        subprocess.check_call([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
                               "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={hsaco}", f"-output={fb}"])
        host = os.path.join(var, f"k_{tag}.o")
        subprocess.check_call(["hipcc", *flags, *inc, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", host])
        lib = os.path.join(var, f"libab_{tag}.so")
        others = [os.path.join(ip.CSRC, f) for f in ("ab_turb_kernels.o", "ab_ice_kernels.o", "ab_phymbl.o", "ab_runtime.o", "ab_sharded.o", "ab_cxx.o")]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, host, *others])
        for f in (s_out, obj, hsaco, fb, host):
            os.remove(f)
        print(f"{mode}: {n} static scalar moves rewritten in {pretty} -> {lib}")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("--kernel", default="flux_kernel<double, 2, true, false, double, double>")
    a = ap.parse_args()
    build(a.kernel)


if __name__ == "__main__":
    main()
