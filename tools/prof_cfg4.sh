#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg4
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o bench -- python3 bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_sq.log 2>&1
python3 tools/rocpd_summary.py $O/pmc_sq/bench_results.db 2>&1 | grep -E "flux_kernel" | cut -c1-170
