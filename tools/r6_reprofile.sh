#!/bin/bash
# Round 6 (GPU box): after the last source change (WL_ECMWF's 1/u*w back to two reciprocals: the shared one spilled five registers of the fp64 ECMWF + skin
# kernel to scratch): A/B against the spilling build, counter profiles of the headline / ECMWF / config 5 / config 2 / config 4 / nb_iter 8 kernels again
# (hash stamp), parity of the ECMWF kernels
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_reprofile
mkdir -p $O
timeout 900 python tools/slab_rates.py --algo ecmwf --rows 3600 --kernels 0 --passes 5 cur spill > $O/ab_ecmwf_spill.txt 2>&1
grep -A3 "^---" $O/ab_ecmwf_spill.txt
bash tools/prof_quick.sh r6_headline
bash tools/prof_quick.sh r6_cfg2 --config 2 --steps 50
bash tools/prof_quick.sh r6_cfg4 --config 4
bash tools/prof_quick.sh r6_cfg5 --config 5 --steps 5
bash tools/prof_quick.sh r6_ecmwf --algo ecmwf
bash tools/prof_quick.sh r6_n8 --niter 8
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_bistable_cells.py tests/test_gpu_mixed.py tests/test_turb_series.py tests/test_skin_modules.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "ecmwf or bistable or skin or mixed or series" > $O/tests.log 2>&1; tail -1 $O/tests.log
python tools/run_configs.py > $O/run_configs.log 2>&1; cp gpurun_out/configs.jsonl $O/configs.jsonl; tail -10 $O/run_configs.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline 2> $O/bench8_d2d.err | grep "^{" > $O/bench8_d2d.json
AEROBULK_AMD_GATHER=rccl python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline 2> $O/bench8_rccl.err | grep "^{" > $O/bench8_rccl.json
