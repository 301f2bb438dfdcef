#!/bin/bash
# Round-6 artefacts in one lease (GPU box): bench line, PMC profiles of the headline and of the config 2 / 4 / 5 / ECMWF / nb_iter 8 kernels,
# every BASELINE configuration, slab rates, host path, the in-process 8-shard runs.  Outputs under gpurun_out/r6_evidence/ and gpurun_out/prof_r6_*/.
R=$GRAFT_REPO_ROOT
cd $R
E=gpurun_out/r6_evidence
mkdir -p $E
python bench.py > $E/bench_final.json 2> $E/bench_final.err
bash tools/prof_quick.sh r6_headline
bash tools/prof_quick.sh r6_cfg2 --config 2 --steps 50
bash tools/prof_quick.sh r6_cfg4 --config 4
bash tools/prof_quick.sh r6_cfg5 --config 5 --steps 5
bash tools/prof_quick.sh r6_ecmwf --algo ecmwf
bash tools/prof_quick.sh r6_n8 --niter 8
python tools/run_configs.py > $E/run_configs.log 2>&1
cp gpurun_out/configs.jsonl $E/configs.jsonl
python tools/slab_rates.py --passes 3 --rows 113,225,450,900,1800,3600 > $E/slab_rates.txt 2>&1
python tools/host_path_bench.py > $E/host_path.txt 2>&1
python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline > $E/bench8_d2d.json 2> $E/bench8_d2d.err
AEROBULK_AMD_GATHER=rccl python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline > $E/bench8_rccl.json 2> $E/bench8_rccl.err
cut -c1-400 $E/bench_final.json; tail -4 $E/run_configs.log; grep -v passes $E/slab_rates.txt | tail -18; tail -5 $E/host_path.txt; cut -c1-300 $E/bench8_rccl.json
