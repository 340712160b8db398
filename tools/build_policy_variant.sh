#!/bin/bash
# A/B build of one policy translation unit: tools/build_policy_variant.sh NAME [x] [-DFLAG ...] -> build/wsdiag/NAME.so
#   (default unit: dpenv_policy.hip; with `x` as second argument: dpenv_policy_x.hip, the fp32-faithful unit)
# The other objects come from the product build in build/obj; select the result with DPENV_LIB=$PWD/build/wsdiag/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
unit=dpenv_policy; others="build/obj/dpenv_policy_x.o"
if [ "$1" = "x" ]; then unit=dpenv_policy_x; others="build/obj/dpenv_policy.o"; shift; fi
mkdir -p /tmp/dpenv_variants build/wsdiag
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -Wno-unused-variable -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DDPENV_BLOCK=64"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -c ml4ca_amd/csrc/$unit.hip -o /tmp/dpenv_variants/${unit}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/wsdiag/$name.so build/obj/dpenv_kernels.o build/obj/dpenv_api.o $others /tmp/dpenv_variants/${unit}_$name.o
echo built build/wsdiag/$name.so
