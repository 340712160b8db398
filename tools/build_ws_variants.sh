#!/bin/bash
# Diagnostic builds of libdpenv.so for the two-wave closed-loop investigation.  Never shipped.
#   build/wsdiag/slp.so              the library with the SLP vectoriser ON in the policy unit, nothing else changed   -> tools/ws_race_check.py
#   build/wsdiag/selfcheck_slp.so    policy unit with the SLP vectoriser ON (packed fp32 in the env wave) + the in-kernel double
#                                    evaluation of env_step (-DDPENV_WS_SELFCHECK) + pk_probe_kernel        -> tools/ws_selfcheck.py
#   build/wsdiag/selfcheck_noslp.so  the same with the product's -fno-slp-vectorize (control: 0 events)      -> tools/ws_pk_probe.py
#   build/wsdiag/sc_<mode>.so        selfcheck_slp with its ISA hand-edited by tools/ws_asm_variant.py (scalarA, swapA, nopBeforeA ...)
#   build/wsdiag/pk_opsel            the stand-alone reproducer tools/pk_opsel_mfma_hazard.hip
# Objects and ISA files go to /tmp/dpenv_variants (they are large and must not travel with the repo snapshot).
set -e
cd "$(dirname "$0")/../ml4ca_amd/csrc"
V=/tmp/dpenv_variants; D=../../build/wsdiag
mkdir -p $V $D
BASE="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DDPENV_BLOCK=64"
hip() { /opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" 2> >(grep -E "error" >&2); }
hip $BASE -fno-slp-vectorize -c dpenv_kernels.hip -o $V/kernels.o &
hip $BASE -fno-slp-vectorize -DDPENV_WS_SELFCHECK -c dpenv_api.hip -o $V/api_sc.o &
hip $BASE -fno-slp-vectorize -c dpenv_api.hip -o $V/api.o &
hip $BASE -c dpenv_policy.hip -o $V/pol_slp.o &
hip $BASE -DDPENV_WS_SELFCHECK -c dpenv_policy.hip -o $V/pol_sc_slp.o &
hip $BASE -fno-slp-vectorize -DDPENV_WS_SELFCHECK -c dpenv_policy.hip -o $V/pol_sc_noslp.o &
hip $BASE -DDPENV_WS_SELFCHECK -S --cuda-device-only dpenv_policy.hip -o $V/pol_sc_slp.s &
wait
hip -shared -o $D/slp.so $V/kernels.o $V/api.o $V/pol_slp.o
hip -shared -o $D/selfcheck_slp.so $V/kernels.o $V/api_sc.o $V/pol_sc_slp.o
hip -shared -o $D/selfcheck_noslp.so $V/kernels.o $V/api_sc.o $V/pol_sc_noslp.o
cd ../..
for m in ${MODES:-identity scalarA swapA nopBeforeA nopAfterA}; do
    python3 tools/ws_asm_variant.py $V/pol_sc_slp.s $V/sc_$m.s $m && tools/ws_asm_link.sh $V/sc_$m.s build/wsdiag/sc_$m.so
done
hip -O3 -fno-slp-vectorize tools/pk_opsel_mfma_hazard.hip -o build/wsdiag/pk_opsel
ls -la build/wsdiag
