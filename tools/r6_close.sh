#!/bin/bash
# Round 6 closing lease (GPU box): ECMWF with zt = zu (psi_h(zt/L) not evaluated) against the previous build, the whole committed GPU suite, then the round's
# artefacts on the closing sources (tools/r6_evidence.sh)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_close
mkdir -p $O
python - > $O/ecmwf_ztzu.txt 2>&1 <<'PY'
import os, subprocess, sys, json
CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import aerobulk_amd as ab
IN6 = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp")
f = ab.synth_fields_device(4320, 3600)
out = {}
for skin in (True, False):
    with ab.Session("ecmwf", 4320, 3600, 1, skin) as s:
        kw = dict(Niter=5, rad_sw=f["rad_sw"] if skin else None, rad_lw=f["rad_lw"] if skin else None, check=False)
        for _ in range(30):
            s.compute(1, 10.0, 10.0, *[f[k] for k in IN6], **kw)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                s.compute(1, 10.0, 10.0, *[f[k] for k in IN6], **kw)
            e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 40)
        out["skin" if skin else "noskin"] = best
print("RESULT", out)
'''
root = os.environ["GRAFT_REPO_ROOT"]
for rep in range(2):
    for tag in ("cur", "prev"):
        e = dict(os.environ)
        if tag != "cur":
            e["AEROBULK_AMD_LIB"] = os.path.join(root, "build", "var", "libab_prev.so")
        p = subprocess.run([sys.executable, "-c", CHILD, root], env=e, capture_output=True, text=True)
        print(tag, [l for l in p.stdout.splitlines() if l.startswith("RESULT")] or p.stderr[-500:])
PY
cat $O/ecmwf_ztzu.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "gputest rc=$?"; tail -2 $O/gputest.log
bash tools/r6_evidence.sh > $O/evidence.log 2>&1; tail -12 $O/evidence.log | cut -c1-250
