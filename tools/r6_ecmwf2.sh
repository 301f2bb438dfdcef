#!/bin/bash
# Round 6 (GPU box), one lease: ECMWF — shared reciprocals, the two UPDATE_QNSOL_TAU calls' common factors hoisted, one reciprocal of u*w in WL_ECMWF —
# against the plain forms (build/var/libab_noshare.so), fp64 and config 5; config 2 with the block kernel's new tail pool; parity of all of it
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_ecmwf2
mkdir -p $O
timeout 1200 python -m pytest tests/test_bistable_cells.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_illcond_cells.py tests/test_gpu_mixed.py tests/test_turb_series.py tests/test_skin_modules.py tests/test_calib.py tests/test_diagnostics.py tests/test_gpu_cu_kernel.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 3600 --kernels 0 --passes 5 cur noshare > $O/ab_ecmwf.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf.txt
one() { python -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["calib"]["fma_f64_tflops_after"], r.get("precision_modes",{}).get("AB_F32",{}).get("value"))'; }
for rep in 1 2; do
  echo "cfg5 cur     $(timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline | one)"
  echo "cfg5 noshare $(AEROBULK_AMD_LIB=$R/build/var/libab_noshare.so timeout 600 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline | one)"
done > $O/cfg5.txt 2>&1; cat $O/cfg5.txt
echo "cfg2 default steps: $(timeout 300 python bench.py --config 2 --no-cpu-baseline | one)" > $O/cfg2.txt 2>&1
echo "cfg2 200 steps:     $(timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline | one)" >> $O/cfg2.txt 2>&1
echo "cfg2 200 steps x1:  $(AEROBULK_AMD_TAIL_X=1 timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline | one)" >> $O/cfg2.txt 2>&1
cat $O/cfg2.txt
AB_TEST_BUDGET_S=0 AB_FUZZ_SEEDS=13100:13120 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "ecmwf" > $O/fuzz_ecmwf.log 2>&1; tail -1 $O/fuzz_ecmwf.log
