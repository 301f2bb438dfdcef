#!/bin/bash
# ECMWF + skin, nb_iter 5 / 8 / 12, with and without the cool skin's g(u) table (build/var/libab_ecsnotab.so: -DAB_ECS_NOTAB): times on one lease
# and the instruction counters of both at nb_iter 5 and 8 (the "not understood" entry of profiles/r4_notes.md section 9)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r5r
python tools/ab_compare.py --passes 3 --configs ecmwf:1:5,ecmwf:1:8,ecmwf:1:12,ecmwf:1:2 cur ecsnotab > gpurun_out/r5r/ab.txt 2>&1
grep -v passes gpurun_out/r5r/ab.txt
bash tools/prof_quick.sh r5_ecmwf_n8 --algo ecmwf --niter 8
export AEROBULK_AMD_LIB=$R/build/var/libab_ecsnotab.so
bash tools/prof_quick.sh r5_ecmwf_notab_n5 --algo ecmwf
bash tools/prof_quick.sh r5_ecmwf_notab_n8 --algo ecmwf --niter 8
