#!/bin/bash
# Round 6, item 1 (GPU box): Ri_bulk without FMA contraction — the four recorded cells, the golden / parity tier, same-lease A/B against round 5's
# library (build/var/libab_r5.so) and a fuzz campaign of the one configuration every rejected cell came from.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_ribulk
mkdir -p $O
timeout 600 python -m pytest tests/test_bistable_cells.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_illcond_cells.py -m gpu -q -x -p no:cacheprovider > $O/tier1.log 2>&1
tail -3 $O/tier1.log
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 3600 --kernels 0 --passes 5 cur r5 > $O/ab_ecmwf.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf.txt
timeout 900 python tools/slab_rates.py --algo coare3p6 --rows 450,3600 --kernels 1 --passes 5 cur r5 > $O/ab_coare.txt 2>&1
grep -A4 "^---" $O/ab_coare.txt
for rng in 5110:5122 9436:9448 11000:11030 11200:11260 12000:12120; do
  AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=$rng timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "corner and ecmwf-True-10.0-10.0-10" > $O/fuzz_$rng.log 2>&1
  echo $rng; tail -1 $O/fuzz_$rng.log
done
