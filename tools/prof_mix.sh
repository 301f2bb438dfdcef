#!/bin/bash
# dynamic instruction mix of the headline kernel at nb_iter = 0 and 5 (fixed part vs iterations): tools/prof_mix.sh <tag>  (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mix_$1
mkdir -p $O
cd $R
for N in 0 1 5; do
ARGS="bench.py --steps 6 --warmup 2 --no-cpu-baseline --niter $N"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/a$N -o bench -- python3 $ARGS > $O/a$N.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/b$N -o bench -- python3 $ARGS > $O/b$N.log 2>&1
echo "#### nb_iter $N"
python3 tools/rocpd_summary.py $O/a$N/bench_results.db $O/b$N/bench_results.db 2>&1 | grep -E "flux_kernel" | cut -c1-60,88-175
done
