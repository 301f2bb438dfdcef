#!/bin/bash
# Round-6 closing soak (GPU box): one full fuzz campaign with new seeds — every configuration of tests/test_gpu_fuzz.py, 60 seeds (~1 400 cases) — plus the
# adversarial fields and the race stress of flux_kernel_cu, on the closing sources
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_soak
mkdir -p $O
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=14000:14060 timeout 3000 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py -m gpu -q -p no:cacheprovider > $O/fuzz_all.log 2>&1
tail -2 $O/fuzz_all.log
timeout 900 python tools/cu_race_stress.py --cases 120 --seed 7 > $O/race.txt 2>&1
tail -2 $O/race.txt
