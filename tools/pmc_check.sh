cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_check
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/micro -o m -- build/var/instr_rate > $O/micro.log 2>&1
python3 tools/rocpd_summary.py $O/micro/m_results.db 2>&1 | cut -c1-220 > $O/micro_summary.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/bench -o b -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
python3 tools/rocpd_summary.py $O/bench/b_results.db 2>&1 | grep -E "^==|flux_kernel|kernel  " | cut -c1-260 > $O/bench_summary.txt
cat $O/micro_summary.txt $O/bench_summary.txt
