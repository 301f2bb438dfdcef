#!/bin/bash
# PMC + stats profile of a bench.py run (GPU box): tools/prof_quick.sh <tag> [bench.py arguments ...]   (default: the headline, config 3)
# separate rocprofv3 passes for the statistics and for every counter group (HBM counters in passes of their own, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd $R
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-nb-iter-8 --no-host-path --no-overlapped $*"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o bench -- python3 $ARGS > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $O/pmc_f64 -o bench -- python3 $ARGS > $O/pmc_f64.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o bench -- python3 $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o bench -- python3 $ARGS > $O/pmc_write.log 2>&1
python3 tools/rocpd_summary.py $O/stats/bench_results.db $O/pmc_sq/bench_results.db $O/pmc_f64/bench_results.db $O/pmc_fetch/bench_results.db $O/pmc_write/bench_results.db 2>&1 | grep -E "^==|flux_kernel|kernel  |-- PMC" | cut -c1-175 > $O/summary.txt
rm -rf $O/stats $O/pmc_sq $O/pmc_f64 $O/pmc_fetch $O/pmc_write      # (the rocpd databases: only the summary travels back)
tail -1 $O/stats.log | cut -c1-300
