#!/bin/bash
# Round 6, item 2 (GPU box): helpers at the end of a launch in flux_kernel_cu (build/var/libab_steal.so, -DAB_CU_STEAL) against the in-tree kernel:
# bits first (race stress, CU-kernel tests), then slab rates, same lease, interleaved passes.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_steal
mkdir -p $O
AEROBULK_AMD_LIB=$R/build/var/libab_steal.so timeout 600 python tools/cu_race_stress.py --cases 80 --seed 6 > $O/race.txt 2>&1
echo "race rc=$?"; tail -3 $O/race.txt
AEROBULK_AMD_LIB=$R/build/var/libab_steal.so timeout 600 python -m pytest tests/test_gpu_cu_kernel.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/tests.log
timeout 1200 python tools/slab_rates.py --rows 225,450,900,1800,3600 --kernels 1 --passes 5 cur steal > $O/slab.txt 2>&1
grep -A7 "^---" $O/slab.txt
