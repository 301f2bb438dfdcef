#!/bin/bash
# Round 6 (GPU box): the committed GPU suite on the closing sources, smoke(), the ECMWF + skin counter profile without the host-path leg's chunk
# kernels in the trace, the in-process eight-shard lines with the peer-slab measurement
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r6_final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "gputest rc=$?"; tail -3 $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
bash tools/prof_quick.sh r6_ecmwf --algo ecmwf > $O/prof_ecmwf.log 2>&1
python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline 2> $O/bench8_d2d.err | grep "^{" > $O/bench8_d2d.json
AEROBULK_AMD_GATHER=rccl python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --verify --no-cpu-baseline 2> $O/bench8_rccl.err | grep "^{" > $O/bench8_rccl.json
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
