#!/bin/bash
# instruction-cache and scalar-data-cache behaviour of the default bench: tools/prof_icache.sh <tag>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd $R
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace -d $O/i1 -o bench -- python3 $ARGS > $O/i1.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/i2 -o bench -- python3 $ARGS > $O/i2.log 2>&1
python3 tools/rocpd_summary.py $O/i1/bench_results.db $O/i2/bench_results.db 2>&1 | grep -E "^==|flux_kernel" | cut -c92-175
