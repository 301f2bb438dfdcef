#!/bin/bash
# Round-5 closing soak (GPU box): the race stress of flux_kernel_cu (first tile by position, counters checked after the launch) and two fuzz
# campaigns with new seeds — every configuration, and the COARE configurations with the CU-wide kernel forced on for the 60 000-cell fields.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r5_soak
timeout 900 python tools/cu_race_stress.py --cases 120 --seed 5 > gpurun_out/r5_soak/race.txt 2>&1
tail -3 gpurun_out/r5_soak/race.txt
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=11000:11030 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py -m gpu -q -p no:cacheprovider > gpurun_out/r5_soak/fuzz_all.log 2>&1
tail -2 gpurun_out/r5_soak/fuzz_all.log
AEROBULK_AMD_CU_KERNEL=1 AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=11100:11116 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py tests/test_gpu_regroup.py -m gpu -q -k coare3p -p no:cacheprovider > gpurun_out/r5_soak/fuzz_cu.log 2>&1
tail -2 gpurun_out/r5_soak/fuzz_cu.log
