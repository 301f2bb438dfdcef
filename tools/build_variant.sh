#!/bin/bash
# build a kernel variant library into build/var/ (travels to the GPU box): tools/build_variant.sh <tag> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=${AB_CONTRACT:-fast-honor-pragmas} -fno-gpu-rdc -Wno-unused-function -mllvm -disable-machine-licm -Xclang -target-feature -Xclang -fmacf64-inst -I include"
/opt/rocm/bin/hipcc $FLAGS "$@" -c aerobulk_amd/csrc/ab_kernels.hip -o build/var/k_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/libab_$tag.so build/var/k_$tag.o aerobulk_amd/csrc/ab_turb_kernels.o aerobulk_amd/csrc/ab_ice_kernels.o aerobulk_amd/csrc/ab_phymbl.o aerobulk_amd/csrc/ab_calib.o aerobulk_amd/csrc/ab_runtime.o aerobulk_amd/csrc/ab_sharded.o aerobulk_amd/csrc/ab_cxx.o
echo build/var/libab_$tag.so
