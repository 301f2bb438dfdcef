#!/bin/bash
# Round 6 (GPU box), one lease: the remaining levers on short launches and small grids, and the ECMWF reciprocal A/B
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_levers
mkdir -p $O
# (a) two kernels, one record
timeout 600 python tools/two_kernel_probe.py > $O/two_kernel.txt 2>&1; grep -v RESULT $O/two_kernel.txt | tail -12
# (c) a grid of flux_kernel_cu that gives every team an integer number of tiles: 4320x450 = 3797 tile-equivalents; 237 workgroups x 4 teams = 948 teams -> 4.0
timeout 900 python tools/slab_rates.py --rows 450 --kernels 1 --passes 3 cur cur@AEROBULK_AMD_CU_GRID=237 cur@AEROBULK_AMD_CU_GRID=240 cur@AEROBULK_AMD_CU_GRID=248 cur@AEROBULK_AMD_CU_GRID=190 > $O/grid.txt 2>&1
grep -A3 "^---" $O/grid.txt | grep -v "^--$" | grep -E "^---|^ +450"
# config 2 (1440x1080 coare3p6 no skin nb_iter 8): tail pool of one-round tiles x0 .. x3, one-round tiles throughout
for x in 1 0 0.5 1.5 2 3; do
  echo "TAIL_X=$x $(AEROBULK_AMD_TAIL_X=$x timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-host-path | python -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["calib"]["fma_f64_tflops_after"])')"
done > $O/cfg2.txt 2>&1
echo "rounds1 $(AEROBULK_AMD_LIB=$R/build/var/libab_rounds1.so timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-host-path | python -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"])')" >> $O/cfg2.txt 2>&1
cat $O/cfg2.txt
# ECMWF: shared reciprocals (cur) against the plain quotients (noshare)
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 3600 --kernels 0 --passes 5 cur noshare > $O/ab_ecmwf.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf.txt
timeout 900 python tools/slab_rates.py --algo ecmwf --niter 8 --rows 3600 --kernels 0 --passes 3 cur noshare > $O/ab_ecmwf8.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf8.txt
# parity of the ECMWF change: golden + parity + bistable + fuzz of the ecmwf configurations
timeout 900 python -m pytest tests/test_bistable_cells.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_illcond_cells.py -m gpu -q -x -p no:cacheprovider > $O/tier1.log 2>&1; tail -1 $O/tier1.log
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=13000:13040 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "ecmwf" > $O/fuzz_ecmwf.log 2>&1; tail -1 $O/fuzz_ecmwf.log
