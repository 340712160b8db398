#!/bin/bash
# A/B build of the two-wave split-arithmetic units: tools/build_xws_variant.sh NAME [-DFLAG ...] -> build/wsdiag/NAME.so
#   dpenv_policy_xws1.hip (all exact) and dpenv_policy_xws2.hip (exact actor) are recompiled with -DDPENV_DEV_FAST (the shipped env /
#   network shape alone: minutes -> ~1 min) plus the given flags; everything else comes from the product build in build/obj.
#   Select with DPENV_LIB=$PWD/build/wsdiag/NAME.so.  Never shipped.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/dpenv_variants build/wsdiag
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -Wno-unused-variable -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DDPENV_BLOCK=64 -DDPENV_DEV_FAST"
# UNITS="ws xws1 xws2" also rebuilds the f16 two-wave unit (dpenv_policy_ws.hip)
objs=""
for u in ws xws1 xws2; do
  if [[ " ${UNITS:-xws1 xws2} " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -c ml4ca_amd/csrc/dpenv_policy_$u.hip -o /tmp/dpenv_variants/${u}_$name.o -Rpass-analysis=kernel-resource-usage 2> /tmp/dpenv_variants/${u}_$name.res &
    objs="$objs /tmp/dpenv_variants/${u}_$name.o"
  else
    objs="$objs build/obj/dpenv_policy_$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/wsdiag/$name.so build/obj/dpenv_kernels.o build/obj/dpenv_api.o build/obj/dpenv_policy.o \
    build/obj/dpenv_policy_x.o $objs
grep -h -A9 "Function Name" /tmp/dpenv_variants/*_$name.res | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: //'
echo built build/wsdiag/$name.so
