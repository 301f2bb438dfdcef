#!/bin/bash
# PC-sampling probe of the default bench (run on the GPU box): tools/prof_pcsamp.sh <tag> [method] [unit] [interval]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pcs_$1
M=${2:-stochastic}; U=${3:-cycles}; I=${4:-65536}
mkdir -p $O
cd $R
rocprofv3 -L > $O/avail.txt 2>&1
grep -n -i -B2 -A12 "pc.sampl" $O/avail.txt | head -60
ARGS="bench.py --steps 30 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace -d $O/run -o bench --output-format csv -- python3 $ARGS > $O/run.log 2>&1
echo "exit $?"; tail -5 $O/run.log
ls -la $O/run
for f in $O/run/*.csv; do echo "== $f"; head -4 $f | cut -c1-400; wc -l $f; done
