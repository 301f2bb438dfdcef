#!/usr/bin/env python
"""Dynamic instruction profile of ONE flux kernel by binary instrumentation (no PC sampling and no thread-trace decoder on this pool).

    python tools/isa_profile.py build  [--kernel 'flux_kernel<double, 2, true, false, double, double>'] [--tag prof]     (CPU box)
    python tools/isa_profile.py run    [--tag prof] [--grid 1440x1080] [--algo coare3p6] [--skin 1] [--niter 5] [--precision f64]  (GPU box)
    python tools/isa_profile.py report [--tag prof] [--counts gpurun_out/prof_counts.json] [-o profiles/r3_valu_breakdown.txt]  (CPU box)

build   compiles aerobulk_amd/csrc/ab_kernels.hip to device assembly (-gline-tables-only: every instruction carries its source line
        and its inlined-at chain), finds the kernel's basic blocks (leaders: labels and the instruction after a branch), and puts
        at the head of every block a sequence that adds 1 per WAVE (also when the wave enters with EXEC = 0) and 1 per ACTIVE LANE to the block's pair of counters in a
        device array (ab_prof_counters, ab_kernels.hip under AB_ISA_PROFILE) with global atomics.  It needs registers the kernel does
        not use: v118-v127 and s[100:101] (asserted from the kernel descriptor), clobbers neither SCC nor VCC, and leaves the kernel's
        results unchanged (`run` checks them against the uninstrumented library).  The patched assembly is assembled, linked,
        bundled and embedded into a host object like hipcc does it, and linked with the other objects into build/var/libab_<tag>.so;
        the static side (blocks, their instructions, classes, source stacks) goes to build/var/<tag>_blocks.json.
run     runs the configuration through that library, reads the counters: gpurun_out/<tag>_counts.json.
report  dynamic wave-instruction counts = sum over blocks of (waves that entered the block) x (instructions of the block), by opcode
        class, by opcode, by innermost source function (qlog, qexp, qdiv ...) and by physics region (the outermost function of
        ab_physics.hpp / ab_tile.hpp on the instruction's inlined-at chain), per cell; lane occupancy per class from the lane counters.
        Cross-check: the total must reproduce rocprofv3's SQ_INSTS_VALU (the -g build differs from the product's by ~0.3 % of its
        instructions).
Issue-slot prices: profiles/r2_instr_rates.txt (one slot = one fp64 FMA of a wave = 4 cycles).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aerobulk_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
SYM = "_ZN2ab16ab_prof_countersE"
NCOUNT = 16384
VBASE = (124, 122, 120, 118)          # VGPR pairs holding counter-window bases; v126 scratch, v127 = 1
WINDOW = 8192                         # bytes addressed by one base pair (13-bit signed offsets)


def hipflags():
    sys.path.insert(0, ROOT)
    from aerobulk_amd import build as b
    return list(b.HIPFLAGS)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


# ---------------------------------------------------------------------------------------------------- source functions
def function_ranges(path):
    """[(first_line, last_line, name)] of the function definitions of a header (brace matching; good enough for these files)."""
    out, depth, base, pending, start = [], 0, None, None, None
    txt = open(path).read().split("\n")
    sig = re.compile(r"(?:__device__|AB_FM|__global__|static\s+inline|inline|static)\b[^;{}]*?\b([A-Za-z_]\w*)\s*\(")
    ns_depth = 0
    cur = None
    for i, line in enumerate(txt, 1):
        code = line.split("//")[0]
        if cur is None:
            m = sig.search(code)
            if m and depth == ns_depth:
                pending = (i, m.group(1))
        for ch in code:
            if ch == "{":
                if cur is None and pending is not None and depth == ns_depth:
                    cur = (pending[0], pending[1], depth)
                    pending = None
                elif cur is None and re.match(r"\s*namespace\b", code) and depth == ns_depth:
                    ns_depth += 1
                depth += 1
            elif ch == "}":
                depth -= 1
                if cur is not None and depth == cur[2]:
                    out.append((cur[0], i, cur[1]))
                    cur = None
                elif cur is None and depth < ns_depth:
                    ns_depth = depth
        if cur is None and pending is not None and ";" in code and "(" in code and "{" not in code:
            pending = None      # a declaration
    return out


class Sources:
    FILES = ("ab_physics.hpp", "ab_fastmath.hpp", "ab_math.hpp", "ab_tile.hpp", "ab_kernels.hip", "ab_physics_ice.hpp")

    def __init__(self):
        self.r = {f: function_ranges(os.path.join(CSRC, f)) for f in self.FILES if os.path.exists(os.path.join(CSRC, f))}

    def func(self, file, line):
        f = os.path.basename(file)
        best = None
        for a, b, name in self.r.get(f, ()):
            if a <= line <= b and (best is None or a >= best[0]):
                best = (a, name)
        if best and best[1] == "__launch_bounds__":       # a __global__ kernel: the parser saw its launch bounds first
            return "flux_kernel"
        return best[1] if best else None


LOC = re.compile(r"^\s*\.loc\s+\d+\s+(\d+)\s+\d+.*?;\s*(\S+?):(\d+):\d+(.*)$")
FRAME = re.compile(r"@\[\s*(\S+?):(\d+):\d+")
# physics regions: the OUTERMOST of these on the inlined-at chain names the region (so that a log inside psi inside first_guess is
# "first_guess_coare", and the iteration's own glue is "turb_coare")
CONTAINERS = {"flux_kernel", "compute_cell", "operator()"}
LEAF_MATH = {"qlog", "qlog10", "qexp", "qexp10", "exp_finish", "qdiv", "qrcp", "qsqrt", "qsqrt_pos", "qrsqrt_pos", "qatan", "qatan_ge1", "qcbrt",
             "qrcbrt_mid", "qrqrt_mid", "horner_coefs", "horner_lit6", "horner_lit4", "psi_tab_eval", "psi_tab_eval32", "log_pair", "ldsc",
             "vconst", "exp_t", "p_fma", "p_rcp", "p_rsq", "p_hi32", "p_lo32", "p_hilo", "p_ldexp", "p_rint", "psi_tab_coef", "horner_tab", "goff_poly"}


def classify(op):
    """(class, issue slots) of a VALU opcode, profiles/r2_instr_rates.txt"""
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "fp64 rcp/rsq/sqrt (quarter rate)", 4.0
    if re.match(r"v_(fma|mul|add|fmac)_f64", op):
        return "fp64 fma/mul/add", 1.0
    if re.match(r"v_(max|min)_f64", op):
        return "fp64 max/min", 1.0
    if re.match(r"v_(ldexp|frexp_mant|frexp_exp_i32|fract|trunc|floor|ceil|rndne)_f64", op):
        return "fp64 ldexp/floor/rndne/frexp", 1.0
    if re.match(r"v_cmpx?_\w+_f64", op):
        return "fp64 compare", 0.75
    if re.match(r"v_cvt_\w*f64|v_cvt_f64", op):
        return "conversions to/from fp64", 1.0
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "fp32 transcendental", 2.0
    if re.match(r"v_cndmask_b32", op):
        return "v_cndmask_b32 (selects)", 1.0
    if re.match(r"v_(readlane|writelane|readfirstlane)_b32", op):
        return "lane moves (SGPR spills, broadcasts)", 1.1
    if re.match(r"v_mov_b64|v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64|v_mad_u64_u32|v_mad_i64_i32|v_add_co|v_addc_co|v_pk_", op):
        return "64-bit moves / integer", 1.0
    if re.match(r"v_mov_b32", op):
        return "v_mov_b32", 0.5
    if re.match(r"v_accvgpr", op):
        return "accvgpr moves", 0.5
    if re.match(r"v_cmpx?_", op):
        return "32-bit compare", 0.5
    if re.match(r"v_(fma|mul|add|sub|mac|fmac|subrev|mad)_f32|v_fmaak_f32|v_fmamk_f32", op):
        return "fp32 fma/mul/add", 0.5
    if re.match(r"v_(max|min|cvt|ldexp|frexp|floor|trunc|rndne|fract)_\w*(f32|i32|u32)|v_cvt_", op):
        return "fp32 max/min/convert (full slot)", 1.0
    if re.match(r"v_(lshlrev_b32|bfi_b32|mul_lo_u32|mul_u32_u24|mul_i32_i24|lshl_or_b32|and_or_b32|add3_u32|perm_b32|lshl_add_u32|add_lshl_u32|bfe_u32|bfe_i32|mad_u32_u24|mbcnt|alignbit|or3_b32|xad_u32|mad_i32_i24)", op):
        return "32-bit integer, full slot (shift-left, 3-operand)", 1.0
    if re.match(r"v_(add|sub|subrev|and|or|xor|not|lshrrev|ashrrev|min|max)_\w*(u32|i32|b32)|v_add_nc", op):
        return "32-bit integer, half slot", 0.5
    return "other VALU", 1.0


def is_terminator(op):
    return op.startswith("s_branch") or op.startswith("s_cbranch") or op.startswith("s_endpgm") or op.startswith("s_setpc") or op.startswith("s_trap")


# ---------------------------------------------------------------------------------------------------- build
def find_kernel(lines, want):
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    dm = demangle([s for _, s in starts])
    for i, s in starts:
        if want in dm[s] and "flux_kernel" in dm[s]:
            j = i
            while not re.match(r"^\.Lfunc_end\d+:", lines[j]):
                j += 1
            return i, j, s, dm[s]
    raise SystemExit(f"no kernel matching {want!r}")


def instrument(lines, i0, i1, mangled):
    """Returns (new lines of [i0, i1), blocks).  blocks[b] = list of (opcode, source stack) of the ORIGINAL instructions."""
    out, blocks, cur, stack = [], [], None, ()
    leader_next = True            # the function entry is a leader

    def counter(b):
        win, off = divmod(16 * b, WINDOW)
        if win >= len(VBASE) or 16 * b + 16 > 4 * NCOUNT:
            raise SystemExit("more blocks than counter windows")
        v = VBASE[win]
        o = off - WINDOW // 2
        # [0] waves: exec forced to lane 0 for one atomic, so that a wave that enters the block with EXEC = 0 (no skip branch: its
        # instructions still issue) is counted; [2] lanes: one atomic per active lane; [1] waves with at least one active lane
        return [f"\ts_mov_b64 s[100:101], exec", f"\ts_mov_b64 exec, 1", f"\tglobal_atomic_add v[{v}:{v + 1}], v127, off offset:{o}",
                f"\ts_mov_b64 exec, s[100:101]", f"\tglobal_atomic_add v[{v}:{v + 1}], v127, off offset:{o + 8}",
                f"\tv_mbcnt_lo_u32_b32 v126, exec_lo, 0", f"\tv_mbcnt_hi_u32_b32 v126, exec_hi, v126", f"\tv_cmp_eq_u32_e64 s[100:101], 0, v126",
                f"\ts_nop 1", f"\tv_cndmask_b32_e64 v126, 0, 1, s[100:101]", f"\tglobal_atomic_add v[{v}:{v + 1}], v126, off offset:{o + 4}",
                f"\ts_waitcnt vmcnt(0)"]

    prologue = [f"\ts_getpc_b64 s[100:101]", f"\ts_add_u32 s100, s100, {SYM}@rel32@lo+4", f"\ts_addc_u32 s101, s101, {SYM}@rel32@hi+12", f"\tv_mov_b32_e32 v127, 1"]
    # one base pair per 8 KB window of counters actually needed (16 B per block): a kernel with up to 512 blocks needs v124-v127 only
    n_lead = sum(1 for ln in lines[i0:i1] if re.match(r"^(\.LBB\d+_\d+):", ln.strip())) + sum(1 for ln in lines[i0:i1] if ln.strip() and is_terminator(ln.strip().split()[0])) + 2
    nwin = min(len(VBASE), -(-16 * n_lead // WINDOW))
    for w, v in enumerate(VBASE[:nwin]):
        prologue += [f"\ts_add_u32 s100, s100, {WINDOW // 2 if w == 0 else WINDOW}", f"\ts_addc_u32 s101, s101, 0", f"\tv_mov_b32_e32 v{v}, s100", f"\tv_mov_b32_e32 v{v + 1}, s101"]
    first = True
    pending_counter = False
    for ln in lines[i0:i1]:
        m = LOC.match(ln)
        if m:
            frames = [(m.group(2), int(m.group(3)))] + [(f, int(l)) for f, l in FRAME.findall(m.group(4))]
            stack = tuple(frames)
            out.append(ln)
            continue
        s = ln.strip()
        if re.match(r"^(\.LBB\d+_\d+|" + re.escape(mangled) + r"):", s):
            leader_next = True
            out.append(ln)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            out.append(ln)
            continue
        op = s.split()[0]
        if leader_next:
            cur = []
            blocks.append(cur)
            if first:
                out += prologue
                first = False
            leader_next = False
            pending_counter = True
        # a join block begins by restoring EXEC (s_or_b64 exec, exec, sN ...): the counters go after those instructions, so that
        # "active lanes" is what the block's own instructions see
        writes_exec = bool(re.match(r"s_\w+\s+exec\b", s)) or "saveexec" in op
        if pending_counter and not writes_exec:
            out += counter(len(blocks) - 1)
            pending_counter = False
        cur.append((op, stack))
        out.append(ln)
        if is_terminator(op):
            if pending_counter:          # a block of exec writes only
                out[-1:-1] = counter(len(blocks) - 1)
                pending_counter = False
            leader_next = True
    return out, blocks, min(VBASE[:nwin])


def cmd_build(a):
    var = os.path.join(ROOT, "build", "var")
    os.makedirs(var, exist_ok=True)
    flags = hipflags()
    src = os.path.join(CSRC, "ab_kernels.hip")
    s_in, s_out = os.path.join(var, f"{a.tag}.s"), os.path.join(var, f"{a.tag}_patched.s")
    inc = ["-I", os.path.join(ROOT, "include")]
    subprocess.check_call(["hipcc", *flags, *inc, "-DAB_ISA_PROFILE", "-gline-tables-only", "--cuda-device-only", "-S", src, "-o", s_in])
    lines = open(s_in).read().split("\n")
    i0, i1, mangled, pretty = find_kernel(lines, a.kernel)
    new, blocks, vmin_used = instrument(lines, i0, i1, mangled)
    text = "\n".join(lines[:i0] + new + lines[i1:])
    # the kernel descriptor and the metadata: the registers the instrumentation uses
    k = re.escape(mangled)
    m = re.search(r"\.amdhsa_kernel " + k + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
    desc = m.group(1)
    nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
    ns = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", desc).group(1))
    assert nv <= vmin_used and ns <= 100, f"kernel uses {nv} VGPRs / {ns} SGPRs: no room for the instrumentation (needs v{vmin_used}-v127, s100-s101)"
    d2 = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 128", desc)
    d2 = re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset 128", d2)
    d2 = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", d2)
    text = text.replace(desc, d2)
    text = re.sub(r"(\.set " + k + r"\.num_vgpr, )\d+", r"\g<1>128", text)
    text = re.sub(r"(\.set " + k + r"\.numbered_sgpr, )\d+", r"\g<1>102", text)
    # YAML metadata of this kernel
    mm = re.search(r"(- \.agpr_count:.*?\.name:\s+" + k + r".*?\.wavefront_size:\s+\d+)", text, re.S)
    if mm:
        blk = mm.group(1)
        # the block found may start at an earlier kernel's entry: cut at the last "- .agpr_count"
        blk = blk[blk.rindex("- .agpr_count"):]
        b2 = re.sub(r"\.vgpr_count:\s+\d+", ".vgpr_count:     128", blk)
        b2 = re.sub(r"\.sgpr_count:\s+\d+", ".sgpr_count:     108", b2)
        text = text.replace(blk, b2)
    open(s_out, "w").write(text)
    os.remove(s_in)
    obj, hsaco, fb = (os.path.join(var, f"{a.tag}.{e}") for e in ("dev.o", "hsaco", "hipfb"))
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_out, "-o", obj])
    subprocess.check_call([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                           "-input=/dev/null", f"-input={hsaco}", f"-output={fb}"])
    host = os.path.join(var, f"k_{a.tag}.o")
    subprocess.check_call(["hipcc", *flags, *inc, "-DAB_ISA_PROFILE", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", host])
    lib = os.path.join(var, f"libab_{a.tag}.so")
    others = [os.path.join(CSRC, f) for f in ("ab_turb_kernels.o", "ab_ice_kernels.o", "ab_phymbl.o", "ab_runtime.o", "ab_sharded.o", "ab_cxx.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, host, *others])
    for f in (s_out, obj, hsaco, fb, host):      # intermediates: tens of MB each, and build/var travels to the GPU box
        os.remove(f)
    src_map = Sources()
    jb = []
    for b in blocks:
        ins = []
        for op, stack in b:
            fr = [(os.path.basename(f), l, src_map.func(f, l)) for f, l in stack]
            ins.append([op, fr])
        jb.append(ins)
    json.dump({"kernel": pretty, "mangled": mangled, "blocks": jb}, open(os.path.join(var, f"{a.tag}_blocks.json"), "w"))
    n_ins = sum(len(b) for b in blocks)
    print(f"{pretty}: {len(blocks)} basic blocks, {n_ins} instructions ({sum(1 for b in blocks for op, _ in b if op.startswith('v_'))} VALU) -> {lib}")


# ---------------------------------------------------------------------------------------------------- run (GPU box)
def cmd_run(a):
    import ctypes as C
    import numpy as np
    lib_path = os.path.join(ROOT, "build", "var", f"libab_{a.tag}.so")
    ni, nj = (int(x) for x in a.grid.split("x"))
    IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
    skin = bool(a.skin)

    def once(lib):
        code = f"""
import os, sys, json, ctypes as C
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
import aerobulk_amd as ab
from aerobulk_amd import _lib
f = ab.synth_fields_device({ni}, {nj}, precision={a.precision!r})
IN6 = {IN6!r}
L = C.CDLL(_lib.LIB_PATH)
prof = hasattr(L, "ab_prof_reset")
with ab.Session({a.algo!r}, {ni}, {nj}, 1, {skin}, precision={a.precision!r}) as s:
    s.set_regroup({a.regroup})
    kw = dict(rad_sw=f["rad_sw"], rad_lw=f["rad_lw"]) if {skin} else {{}}
    o = s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter={a.niter}, **kw)
    if prof:
        assert L.ab_prof_reset() == 0
    o = s.compute(1, 2.0, 10.0, *[f[k] for k in IN6], Niter={a.niter}, **kw)
    torch.cuda.synchronize()
    cnt = None
    if prof:
        buf = (C.c_uint * {NCOUNT})()
        assert L.ab_prof_read(buf, {NCOUNT}) == 0
        cnt = list(buf)
    sums = {{k: float(v.double().sum()) for k, v in o.items()}}
    h = {{k: v.cpu().numpy().tobytes().hex()[:0] for k, v in o.items()}}
    import hashlib
    dig = {{k: hashlib.sha1(v.cpu().numpy().tobytes()).hexdigest() for k, v in o.items()}}
print("RESULT " + json.dumps(dict(counts=cnt, sums=sums, digest=dig)))
"""
        env = dict(os.environ)
        if lib:
            env["AEROBULK_AMD_LIB"] = lib
        else:
            env.pop("AEROBULK_AMD_LIB", None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        for ln in out.stdout.splitlines():
            if ln.startswith("RESULT "):
                return json.loads(ln[7:])
        raise SystemExit(out.stdout[-2000:] + out.stderr[-4000:])

    ref = once(None)
    got = once(lib_path)
    same = ref["digest"] == got["digest"]
    print("outputs of the instrumented kernel bit-identical to the product's:", same)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    meta = dict(grid=[ni, nj], algo=a.algo, skin=skin, niter=a.niter, precision=a.precision, regroup=a.regroup, identical=same)
    json.dump(dict(meta=meta, counts=got["counts"]), open(os.path.join(ROOT, "gpurun_out", f"{a.tag}_counts.json"), "w"))
    if not same:
        raise SystemExit("instrumented kernel changed the results")


# ---------------------------------------------------------------------------------------------------- report
def region_of(frames):
    """Physics region: the outermost named function on the chain that is not a container and not a math leaf."""
    names = [fn for _, _, fn in frames if fn]
    for fn in reversed(names):                      # frames are innermost first
        if fn in CONTAINERS or fn in LEAF_MATH:
            continue
        return fn
    for fn in reversed(names):
        if fn not in CONTAINERS:
            return fn
    return names[-1] if names else "?"


def second_region(frames):
    """One level below the region (e.g. turb_coare -> cool_skin)."""
    names = [fn for _, _, fn in frames if fn and fn not in CONTAINERS and fn not in LEAF_MATH]
    names = list(reversed(names))
    return " > ".join(names[:2]) if names else "?"


def leaf_of(frames):
    for _, _, fn in frames:
        if fn:
            return fn
    return "?"


def cmd_report(a):
    var = os.path.join(ROOT, "build", "var")
    st = json.load(open(os.path.join(var, f"{a.tag}_blocks.json")))
    cj = json.load(open(a.counts or os.path.join(ROOT, "gpurun_out", f"{a.tag}_counts.json")))
    counts, meta = cj["counts"], cj["meta"]
    cells = meta["grid"][0] * meta["grid"][1]
    blocks = st["blocks"]
    by_class, by_op, by_region, by_leaf, by_path = (collections.Counter() for _ in range(5))
    slots_class, lanes_class = collections.Counter(), collections.Counter()
    region_class = collections.defaultdict(collections.Counter)
    by_line, line_class, line_fn = collections.Counter(), collections.defaultdict(collections.Counter), {}
    tot_valu = tot_salu = tot_lds = tot_vmem = tot_smem = tot_other = 0
    exec0_valu, exec0_blocks = 0, []
    for b, ins in enumerate(blocks):
        w, wnz, ln = counts[4 * b], counts[4 * b + 1], counts[4 * b + 2]
        if not w:
            continue
        nv = sum(1 for op, _ in ins if op.startswith("v_"))
        exec0_valu += (w - wnz) * nv
        if w - wnz and nv:
            exec0_blocks.append(((w - wnz) * nv, b, w, wnz, nv, second_region(next(fr for op, fr in ins if op.startswith("v_")))))
        for op, fr in ins:
            if op.startswith("v_"):
                c, price = classify(op)
                by_class[c] += w
                slots_class[c] += w * price
                lanes_class[c] += ln
                by_op[op] += w
                r = region_of(fr)
                by_region[r] += w
                region_class[r][c] += w
                by_leaf[leaf_of(fr)] += w
                by_path[second_region(fr)] += w
                # the source line of the OUTERMOST physics frame below the math leaves: where in ab_physics.hpp the instruction was asked for
                phys = [(f_, l_, fn) for f_, l_, fn in fr if fn and fn not in LEAF_MATH and fn not in CONTAINERS and l_]
                key = (phys[0][0], phys[0][1]) if phys else (fr[0][0], fr[0][1])
                by_line[key] += w
                line_class[key][c] += w
                line_fn[key] = phys[0][2] if phys else leaf_of(fr)
                tot_valu += w
            elif op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop", "s_barrier", "s_store", "s_atomic", "s_dcache")):
                tot_salu += w
            elif op.startswith("ds_"):
                tot_lds += w
            elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
                tot_vmem += w
            elif op.startswith(("s_load", "s_buffer")):
                tot_smem += w
            else:
                tot_other += w
    per = 64.0 / cells            # wave-instructions -> lane-instructions per cell (the unit of SQ_INSTS_VALU x 64 / cells)
    L = []
    p = L.append
    p(f"tools/isa_profile.py: dynamic instruction profile of {st['kernel']}")
    p(f"{meta['algo']} skin={int(meta['skin'])} nb_iter={meta['niter']} {meta['precision']} regroup={meta['regroup']} on {meta['grid'][0]}x{meta['grid'][1]} synthetic cells (MI355X); "
      f"every basic block counted by an instrumented build ({len(blocks)} blocks; results bit-identical to the product's: {meta['identical']}).")
    p("Unit: VALU instructions per cell = wave-instructions x 64 / cells, the unit of rocprofv3's SQ_INSTS_VALU x 64 / cells.  Issue slots: profiles/r2_instr_rates.txt")
    p("(one slot = 4 cycles = one fp64 FMA of a wave).  The -gline-tables-only build differs from the product's by ~0.3 % of its instructions.")
    p("")
    p(f"VALU {tot_valu * per:8.1f} per cell   (SALU {tot_salu * per:7.1f}, LDS {tot_lds * per:6.1f}, SMEM {tot_smem * per:5.1f}, VMEM {tot_vmem * per:5.1f}, waits/nops/barriers/branches not listed)")
    p(f"VALU instructions issued by waves with NO active lane in the block (entered with EXEC = 0, no skip branch): {exec0_valu * per:.1f} per cell "
      f"({100.0 * exec0_valu / tot_valu:.1f} %)")
    for d, b, w, wnz, nv, reg in sorted(exec0_blocks, reverse=True)[:12]:
        if d * per >= 1.0:
            p(f"      block {b:4d}: {w - wnz:9d} of {w:9d} entries with EXEC = 0, {nv:3d} VALU instructions -> {d * per:6.1f} per cell   [{reg}]")
    p("")
    p("== by opcode class:  instructions per cell | share | issue slots per cell | lanes active")
    fp64 = ("fp64 fma/mul/add", "fp64 rcp/rsq/sqrt (quarter rate)")
    for c, w in by_class.most_common():
        p(f"   {c:52s} {w * per:8.1f}  {100.0 * w / tot_valu:5.1f} %  {slots_class[c] * per:8.1f}   {100.0 * lanes_class[c] / (64.0 * w):5.1f} %")
    nf = sum(w for c, w in by_class.items() if c not in fp64)
    p(f"   -> fp64 arithmetic {sum(by_class[c] for c in fp64) * per:.1f} per cell; everything else {nf * per:.1f} per cell ({100.0 * nf / tot_valu:.1f} %)")
    p(f"   -> issue slots {sum(slots_class.values()) * per:.1f} per cell")
    p("")
    p("== the 40 most executed opcodes (instructions per cell)")
    for op, w in by_op.most_common(40):
        p(f"   {op:28s} {w * per:8.1f}  {100.0 * w / tot_valu:5.1f} %   [{classify(op)[0]}]")
    p("")
    p("== by physics region (outermost function of the inlined-at chain below the kernel / compute_cell): instructions per cell, of which not fp64 arithmetic")
    for r, w in by_region.most_common():
        other = sum(v for c, v in region_class[r].items() if c not in fp64)
        p(f"   {r:28s} {w * per:8.1f}  {100.0 * w / tot_valu:5.1f} %   non-fp64 {other * per:7.1f}")
    p("")
    p("== by call path, two levels (instructions per cell)")
    for r, w in by_path.most_common(40):
        p(f"   {r:56s} {w * per:8.1f}  {100.0 * w / tot_valu:5.1f} %")
    p("")
    p("== by innermost source function (instructions per cell)")
    for r, w in by_leaf.most_common(40):
        p(f"   {r:28s} {w * per:8.1f}  {100.0 * w / tot_valu:5.1f} %")
    p("")
    p("== what is NOT fp64 arithmetic, by region and class (instructions per cell; rows above 8 per cell)")
    rows = []
    for r, cc in region_class.items():
        for c, w in cc.items():
            if c not in fp64 and w * per >= 8.0:
                rows.append((w, r, c))
    for w, r, c in sorted(rows, reverse=True):
        p(f"   {r:28s} {c:52s} {w * per:8.1f}")
    p("")
    p("== by source line of the innermost physics function (instructions per cell; of which not fp64 arithmetic; the classes above 2 per cell)")
    for key, w in by_line.most_common(70):
        other = sum(v for c, v in line_class[key].items() if c not in fp64)
        det = ", ".join(f"{c.split(' (')[0]} {v * per:.0f}" for c, v in line_class[key].most_common() if c not in fp64 and v * per >= 2.0)
        p(f"   {key[0]}:{key[1]:<5d} {line_fn[key]:22s} {w * per:7.1f}   non-fp64 {other * per:6.1f}   {det}")
    txt = "\n".join(L) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("--kernel", default="flux_kernel<double, 2, true, false, double, double>")
    b.add_argument("--tag", default="prof")
    r = sub.add_parser("run")
    r.add_argument("--tag", default="prof")
    r.add_argument("--grid", default="1440x1080")
    r.add_argument("--algo", default="coare3p6")
    r.add_argument("--skin", type=int, default=1)
    r.add_argument("--niter", type=int, default=5)
    r.add_argument("--precision", default="f64")
    r.add_argument("--regroup", type=int, default=1)
    p = sub.add_parser("report")
    p.add_argument("--tag", default="prof")
    p.add_argument("--counts", default=None)
    p.add_argument("-o", "--out", default=None)
    a = ap.parse_args()
    {"build": cmd_build, "run": cmd_run, "report": cmd_report}[a.cmd](a)


if __name__ == "__main__":
    main()
