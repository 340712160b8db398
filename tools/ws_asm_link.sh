#!/bin/bash
# ws_asm_link.sh edited.s out.so [api object]: assemble an edited device ISA of dpenv_policy.hip, embed it in the host object
# and link a diagnostic libdpenv variant (objects of the other two units from tools/build_ws_variants.sh in /tmp/dpenv_variants)
set -e
L=/opt/rocm/lib/llvm/bin; V=/tmp/dpenv_variants; S=$(realpath "$1"); O=$(realpath -m "$2"); API=${3:-$V/api_sc.o}
cd "$(dirname "$0")/../ml4ca_amd/csrc"
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$S" -o "$S.dev.o"
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$S.out" "$S.dev.o"
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input="$S.out" -output="$S.hipfb"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDPENV_BLOCK=64 -DDPENV_WS_SELFCHECK --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$S.hipfb" -c dpenv_policy.hip -o "$S.host.o" 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$O" $V/kernels.o $API "$S.host.o"
