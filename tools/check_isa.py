#!/usr/bin/env python3
"""Hard build step of libdpenv.so (ml4ca_amd/csrc/Makefile): the LINKED library - every kernel instantiation of every translation
unit - must contain no packed fp32 arithmetic.

Why: one form of it, ``v_pk_fma_f32 ... op_sel:[0,1,0]`` / ``[0,0,1]``, now and then loses its product term in lanes 48-63 on MI355X
while another wave of the same SIMD has VALU work in the shadow of its MFMAs (tools/pk_opsel_mfma_hazard.hip) -
which is exactly how the two-wave closed-loop kernels run.  The library is built with -fno-slp-vectorize so that the form cannot
appear; this makes a build that brings it back (a changed flag, a new compiler default, a hand-written packed op) FAIL instead of
shipping.  The f16 packed forms (v_pk_mul_f16, v_pk_max_f16: the network's activation packing) are fine.

    tools/check_isa.py path/to/libdpenv.so      exit status 1 and the offending instructions if anything is found
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile



def find_objdump():
    """llvm-objdump of the ROCm install in use: $ROCM_PATH, the prefix `hipconfig --rocmpath` reports, /opt/rocm, then PATH"""
    cands = []
    if os.environ.get('ROCM_PATH'):
        cands.append(os.path.join(os.environ['ROCM_PATH'], 'lib', 'llvm', 'bin', 'llvm-objdump'))
    try:
        root = subprocess.run(['hipconfig', '--rocmpath'], capture_output=True, text=True, timeout=20).stdout.strip()
        if root:
            cands.append(os.path.join(root, 'lib', 'llvm', 'bin', 'llvm-objdump'))
    except Exception:
        pass
    cands.append('/opt/rocm/lib/llvm/bin/llvm-objdump')
    for c in cands:
        if os.path.exists(c):
            return c
    w = shutil.which('llvm-objdump')
    if w:
        return w
    raise SystemExit('check_isa: no llvm-objdump found (ROCM_PATH, hipconfig --rocmpath, /opt/rocm, PATH)')


OBJDUMP = find_objdump()


def device_disassembly(lib):
    """{code object name: disassembly} of every gfx9xx code object bundled in `lib`"""
    tmp = tempfile.mkdtemp(prefix='dpenv_isa_')
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([OBJDUMP, '--offloading', work], check=True, capture_output=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if 'amdgcn' in f:
                out[f] = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    lib = sys.argv[1]
    if not os.path.exists(OBJDUMP):
        print('check_isa: %s not found: cannot verify the library' % OBJDUMP)
        return 1
    dis = device_disassembly(lib)
    if not dis:
        print('check_isa: no device code objects found in %s' % lib)
        return 1
    bad, n_inst, n_mfma = [], 0, 0
    for name, txt in dis.items():
        for ln in txt.splitlines():
            m = re.search(r'\b(v_pk_\w+)\b', ln)
            n_inst += 1
            n_mfma += 'v_mfma' in ln
            if m and not m.group(1).endswith('_f16'):
                bad.append((name, ln.strip()))
    if bad:
        print('check_isa: %d packed non-f16 instructions in %s (built without -fno-slp-vectorize?):' % (len(bad), lib))
        for b in bad[:10]:
            print('   ', b)
        return 1
    if n_mfma == 0:
        print('check_isa: no MFMA found - is this libdpenv.so?')
        return 1
    print('check_isa: %d code objects, %d lines of disassembly, %d MFMAs, no packed fp32 arithmetic' % (len(dis), n_inst, n_mfma))
    return 0


if __name__ == '__main__':
    sys.exit(main())
