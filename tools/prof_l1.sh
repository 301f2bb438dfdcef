#!/bin/bash
# L1 / texture-addresser counters of the flux kernels (run on the GPU box): tools/prof_l1.sh <tag> <bench.py args...>
# (few counters per pass: a pass that asks for more TA / TCP counters than the hardware has aborts inside rocprofv3 and then hangs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/prof_l1_$tag
mkdir -p $O
cd $R
ARGS="bench.py $* --steps 6 --warmup 2 --no-cpu-baseline"
timeout 240 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -o b -- python3 $ARGS > $O/p1.log 2>&1
timeout 240 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-trace -d $O/p2 -o b -- python3 $ARGS > $O/p2.log 2>&1
timeout 240 rocprofv3 --pmc TCP_PERF_SEL_TOTAL_READ TCP_TCC_READ_REQ_sum --kernel-trace -d $O/p3 -o b -- python3 $ARGS > $O/p3.log 2>&1
timeout 240 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace -d $O/p4 -o b -- python3 $ARGS > $O/p4.log 2>&1
python3 tools/rocpd_summary.py $O/p1/b_results.db $O/p2/b_results.db $O/p3/b_results.db $O/p4/b_results.db 2>&1 | grep -E "flux_kernel" | cut -c1-170
