#!/bin/bash
# Round-1 profiling recipe (run on the GPU box via gpurun). Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1
mkdir -p $O
cd $R
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o bench -- python3 $ARGS > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o bench -- python3 $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o bench -- python3 $ARGS > $O/pmc_write.log 2>&1
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|VALUBusy|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU_FMA_F64|SQ_INSTS_VALU_ADD_F64|SQ_INSTS_VALU_MUL_F64|SQ_INSTS_VALU_TRANS_F64)" | head -20 > $O/counters.txt
rocprofv3 -L 2>/dev/null | grep -i -E "F64|TRANS" | head -40 >> $O/counters.txt
find $O -name "*.csv" | head -30
tail -2 $O/stats.log
