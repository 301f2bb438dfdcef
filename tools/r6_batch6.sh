#!/bin/bash
# Round 6 (GPU box), one lease: COARE's two quotients from one reciprocal (cur) against the two quotients (nocs) and round 5's library (r5) on the
# headline and on the 450-row slab; ECMWF + skin against r5; config 5; parity on every cell of the benchmark grid for the configurations whose arithmetic changed
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_batch6
mkdir -p $O
timeout 1200 python tools/slab_rates.py --rows 450,3600 --kernels 1 --passes 5 cur nocs r5 > $O/ab_coare.txt 2>&1
grep -A4 "^---" $O/ab_coare.txt
timeout 900 python tools/slab_rates.py --rows 3600 --kernels 0 --passes 3 cur nocs > $O/ab_coare_block.txt 2>&1
grep -A3 "^---" $O/ab_coare_block.txt
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 3600 --kernels 0 --passes 5 cur noshare r5 > $O/ab_ecmwf.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf.txt
one() { python -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["value_norm"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["calib"]["fma_f64_tflops_after"], r.get("precision_modes",{}).get("AB_F32",{}).get("value"))'; }
for rep in 1 2; do
  echo "cfg5 cur     $(timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline | one)"
  echo "cfg5 r5      $(AEROBULK_AMD_LIB=$R/build/var/libab_r5.so timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline | one)"
done > $O/cfg5.txt 2>&1; cat $O/cfg5.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_bistable_cells.py tests/test_illcond_cells.py tests/test_gpu_mixed.py tests/test_calib.py tests/test_gpu_adversarial.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=13200:13212 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
