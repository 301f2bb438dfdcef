#!/bin/bash
# Round-6 second closing soak (GPU box): 60 more new seeds of every configuration of tests/test_gpu_fuzz.py, on the closing sources; keeps going after a failure
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_soak2
mkdir -p $O
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=15000:15060 timeout 3300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $O/fuzz_all.log 2>&1
tail -4 $O/fuzz_all.log
grep -E "^FAILED" $O/fuzz_all.log | head
