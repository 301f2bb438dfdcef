#!/bin/bash
# Round 6 (GPU box): half of the block kernel's one-round tiles at the FRONT of the grid, mixed with two-round tiles (build/var/libab_front.so, -DAB_FRONT_SHORT), config 2 and slabs
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_front
mkdir -p $O
one() { python -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["calib"]["fma_f64_tflops_after"])'; }
for rep in 1 2 3; do
  echo "cfg2 cur   $(timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline | one)"
  echo "cfg2 front $(AEROBULK_AMD_LIB=$R/build/var/libab_front.so timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline | one)"
done > $O/cfg2.txt 2>&1; cat $O/cfg2.txt
timeout 900 python tools/slab_rates.py --rows 225,450,3600 --kernels 0 --passes 3 cur front > $O/slab.txt 2>&1; grep -A5 "^---" $O/slab.txt
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 450,3600 --kernels 0 --passes 3 cur front > $O/slab_ecmwf.txt 2>&1; grep -A4 "^---" $O/slab_ecmwf.txt
AEROBULK_AMD_LIB=$R/build/var/libab_front.so timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_regroup.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -1 $O/tests.log
