#!/usr/bin/env python3
"""Static instruction mix of the kernels in the ISA that `make -C ml4ca_amd/csrc asm` keeps under build/asm/.

    python tools/isa_count.py [substring of the mangled kernel name ...]

At 65 536 envs every SIMD holds ONE wave, and a lone wave issues about one instruction per 5-6 cycles whatever it is, so
instructions per env-step per wave is the currency the env kernels are tuned in.  Counts are
static (whole kernel body, all paths); the sub-step loop is unrolled x10 and runs twice per env step.
"""
import collections
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'MFMA'
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_endpgm'):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'SMEM'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'VMEM'
    return 'other'


def main():
    pats = sys.argv[1:]
    for path in sorted(glob.glob(os.path.join(ROOT, 'build', 'asm', '*gfx950.s'))):
        name, counts = None, None
        for line in open(path):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                name, counts = m.group(1), collections.Counter()
                continue
            if name is None:
                continue
            t = line.strip()
            if t.startswith('.end_amdhsa_kernel') or t.startswith('.section') or t.startswith('.Lfunc_end'):
                if counts and (not pats or any(p in name for p in pats)):
                    tot = sum(counts.values())
                    print('%-78s %5d instr: %s' % (name[:78], tot, '  '.join('%s %d' % kv for kv in sorted(counts.items()))))
                name = None
                continue
            if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
                continue
            counts[classify(t.split()[0])] += 1


if __name__ == '__main__':
    main()
