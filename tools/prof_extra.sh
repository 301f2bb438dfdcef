#!/bin/bash
# extra PMC passes of the default bench (what keeps the VALU from 100 %?): tools/prof_extra.sh <tag>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd $R
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM --kernel-trace -d $O/x1 -o bench -- python3 $ARGS > $O/x1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace -d $O/x2 -o bench -- python3 $ARGS > $O/x2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/x3 -o bench -- python3 $ARGS > $O/x3.log 2>&1
python3 tools/rocpd_summary.py $O/x1/bench_results.db $O/x2/bench_results.db $O/x3/bench_results.db 2>&1 | grep -E "^==|flux_kernel" | cut -c92-175
