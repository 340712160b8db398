#!/bin/bash
# A/B or diagnostic build of the env translation unit: tools/build_kernels_variant.sh NAME [BLOCK=n] [-DFLAG ...] -> build/wsdiag/NAME.so
#   dpenv_kernels.hip (step / rollout / reset / GAE kernels) is recompiled with the given flags; the policy units and the host side come
#   from the product build in build/obj (run `make -C ml4ca_amd/csrc` first).  Select the result with DPENV_LIB=$PWD/build/wsdiag/NAME.so.
#   Never shipped: the product library is ml4ca_amd/lib/libdpenv.so from the Makefile alone.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
block=64
if [[ "$1" == BLOCK=* ]]; then block=${1#BLOCK=}; shift; fi
mkdir -p /tmp/dpenv_variants build/wsdiag
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -Wno-unused-variable -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DDPENV_BLOCK=$block"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -c ml4ca_amd/csrc/dpenv_kernels.hip -o /tmp/dpenv_variants/kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/wsdiag/$name.so /tmp/dpenv_variants/kernels_$name.o build/obj/dpenv_api.o \
    build/obj/dpenv_policy.o build/obj/dpenv_policy_ws.o build/obj/dpenv_policy_x.o build/obj/dpenv_policy_xws1.o build/obj/dpenv_policy_xws2.o
echo built build/wsdiag/$name.so
