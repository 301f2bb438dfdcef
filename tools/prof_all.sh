#!/bin/bash
# per-kernel durations of everything the library launches (run on the GPU box): the BASELINE configurations through bench.py, the
# TURB_* / sea-ice / neutral-10 m boundaries through their GPU tests.  tools/prof_all.sh <tag>  ->  gpurun_out/prof_<tag>/all.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/configs -o all -- python3 tools/run_configs.py > $O/configs.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/bound -o all -- python3 -m pytest tests/test_sea_ice.py tests/test_turb_series.py tests/test_neutral10.py tests/test_diagnostics.py -q -m gpu -k "not fortran" > $O/bound.log 2>&1
python3 tools/rocpd_summary.py $O/configs/all_results.db $O/bound/all_results.db 2>&1 | cut -c1-190 > $O/all.txt
tail -5 $O/configs.log; tail -2 $O/bound.log; wc -l $O/all.txt
