#!/usr/bin/env python
"""Static census of one kernel's gfx950 ISA: instruction classes per loop nest.

    hipcc <flags of aerobulk_amd/build.py> -S --cuda-device-only aerobulk_amd/csrc/ab_kernels.hip -o build/asm/k.s
    python tools/isa_census.py build/asm/k.s 'flux_kernel<double, 1, true, false, double>'

Loops are found from backward branches (a branch to a label defined earlier); every instruction is attributed to the innermost
loop that contains its line.  Issue-slot weights as measured in profiles/r1_instr_rates.txt (fp64 arithmetic 1, v_rcp/rsq/sqrt_f64
4, 32-bit VALU 0.5, fp32 transcendentals 2).  This is a static count: a loop's body is weighted once, whatever its trip count.
"""
import collections
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernel_lines(path, want):
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    dm = demangle([s for _, s in starts])
    for k, (i, s) in enumerate(starts):
        if want in dm[s]:
            end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm")), len(lines))
            return lines[i:end + 1], dm[s]
    raise SystemExit(f"no kernel matching {want!r}")


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            return "fp64 rcp/rsq", 4.0
        if re.match(r"v_(fma|mul|add|fmac)_f64", op):
            return "fp64 fma/mul/add", 1.0
        if re.match(r"v_(max|min|ldexp|frexp_mant|fract|trunc|floor|ceil|rndne|div_fixup|div_scale|div_fmas)_f64", op) or op == "v_max_num_f64":
            return "fp64 other (max/min/ldexp..)", 1.0
        if re.match(r"v_cmp\w*_f64", op) or re.match(r"v_cmpx\w*_f64", op):
            return "fp64 compare", 1.0
        if re.match(r"v_cvt_\w*f64|v_cvt_f64", op) or "f64" in op:
            return "fp64 convert/other", 1.0
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op):
            return "fp32 transcendental", 2.0
        if op.startswith("v_mov_b64") or op.startswith("v_lshl_add_u64") or "u64" in op or "i64" in op or "b64" in op:
            return "64-bit int/move", 1.0
        if op.startswith("v_mov_b32"):
            return "v_mov_b32", 0.5
        if op.startswith("v_cndmask"):
            return "v_cndmask_b32", 0.5
        if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
            return "lane moves (spills, broadcasts)", 0.5
        if op.startswith("v_accvgpr"):
            return "accvgpr moves", 0.5
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "32-bit compare", 0.5
        return "32-bit other", 0.5
    if op.startswith("ds_"):
        return "LDS", 0.0
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM", 0.0
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM", 0.0
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait/nop/barrier", 0.0
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch", 0.0
    if op.startswith("s_"):
        return "SALU", 0.0
    return "other", 0.0


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines, name = kernel_lines(path, want)
    labels = {}
    insts = []   # (line index, op, operands)
    for i, l in enumerate(lines):
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            if (m := re.match(r"^(\.LBB\w+):", s)):
                labels[m.group(1)] = i
            continue
        if (m := re.match(r"^(\.?\w+):", s)):
            labels[m.group(1)] = i
            continue
        parts = s.split(None, 1)
        insts.append((i, parts[0], parts[1] if len(parts) > 1 else ""))
    loops = []   # (start, end)
    for i, op, args in insts:
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = args.strip()
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i))
    loops = sorted(set(loops))

    def innermost(i):
        best = None
        for a, b in loops:
            if a <= i <= b and (best is None or (b - a) < (best[1] - best[0])):
                best = (a, b)
        return best

    per = collections.defaultdict(lambda: collections.Counter())
    slots = collections.defaultdict(float)
    for i, op, _ in insts:
        lp = innermost(i)
        c, w = classify(op)
        per[lp][c] += 1
        slots[lp] += w
    print(name)
    print(f"{len(insts)} instructions, {len(loops)} loops")
    for lp in sorted(per, key=lambda x: (-1, -1) if x is None else x):
        cnt = per[lp]
        valu = sum(v for k, v in cnt.items() if k.startswith(("fp64", "fp32", "64-bit", "v_", "lane", "accvgpr", "32-bit")))
        depth = 0 if lp is None else sum(1 for a, b in loops if a <= lp[0] and lp[1] <= b)
        tag = "outside loops" if lp is None else f"loop lines {lp[0]}-{lp[1]} (depth {depth})"
        print(f"\n== {tag}: {sum(cnt.values())} instructions, {valu} VALU, {slots[lp]:.0f} issue slots")
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
            print(f"   {k:34s} {v:6d}")
    if len(sys.argv) > 3:   # dump mnemonic histogram of the loop containing a given line
        tgt = int(sys.argv[3])
        lp = innermost(tgt)
        h = collections.Counter(op for i, op, _ in insts if innermost(i) == lp)
        print(f"\nmnemonics of loop {lp}:")
        for k, v in h.most_common(60):
            print(f"   {k:28s} {v}")


if __name__ == "__main__":
    main()
