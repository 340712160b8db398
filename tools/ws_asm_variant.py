#!/usr/bin/env python3
"""Hand-edit the ISA of the policy translation unit (hipcc -S --cuda-device-only output) for the two-wave investigation.
   ws_asm_variant.py in.s out.s MODE
MODE: scalarA  - every  v_pk_fma_f32 D, S0, S1, S2 op_sel:[0,1,0]  (low result multiplies by S1's HIGH register) becomes two v_fma_f32
      swapA    - the same instruction with src0/src1 exchanged (the lo <- hi swizzle then sits on src0): op_sel:[1,0,0]
      nopBeforeA / nopAfterA - s_nop 7 in front of / behind that instruction
      identity - no change (pipeline check)"""
import re, sys
src, dst, mode = sys.argv[1:4]
pat = re.compile(r'^\s+v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]\s*$')
out, n = [], 0
for ln in open(src).read().split('\n'):
    m = pat.match(ln)
    if m and mode != 'identity':
        d0, d1, a0, a1, b0, b1, c0, c1 = map(int, m.groups())
        n += 1
        if mode == 'scalarA':
            out.append('\tv_fma_f32 v%d, v%d, v%d, v%d' % (d0, a0, b1, c0))
            out.append('\tv_fma_f32 v%d, v%d, v%d, v%d' % (d1, a1, b1, c1))
        elif mode == 'swapA':
            out.append('\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0,0]' % (d0, d1, b0, b1, a0, a1, c0, c1))
        elif mode == 'nopBeforeA':
            out.append('\ts_nop 7'); out.append(ln)          # 8 wait states between the producer of src1 and the swizzled read
        elif mode == 'nopAfterA':
            out.append(ln); out.append('\ts_nop 7')
        else:
            raise SystemExit('unknown mode')
    else:
        out.append(ln)
open(dst, 'w').write('\n'.join(out))
print(mode, 'edited', n, 'instructions')
