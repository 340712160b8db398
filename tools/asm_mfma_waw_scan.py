#!/usr/bin/env python3
"""Static check of gfx950 ISA (hipcc -S output): does an instruction INSIDE an inline-asm statement write a VGPR that lies in the
destination tile of an MFMA which may still be in flight?

The compiler's MFMA hazard recogniser does not look inside asm statements.  Reads are covered in the sources by a compiler-visible
VALU read in front of the asm (dpenv_policy_dev.h, HAZARD NOTE); this checks the WRITE side: registers of an accumulator tile that
are never read afterwards (rows 80..95 of an 80-wide layer) are dead for the register allocator as soon as the tile's MFMAs are
issued, and it may hand them to the next asm statement as temporaries - which the still-running MFMA then overwrites.  Rule: no
instruction between ;;#ASMSTART and ;;#ASMEND may write into the destination of one of the `window` most recent MFMAs issued
at most `max_dist` instructions earlier, unless a compiler-visible VALU instruction has read that MFMA's result (or a younger
MFMA's: the matrix pipe is in order) in between - such a read carries the recogniser's wait states.  The scan is linear in the
file; the window is dropped after an unconditional branch (the textually next block is then not the successor).
The READ side is checked the same way (round 3, asm_reads_of_unlanded_mfma_dest): an asm instruction that reads a register of
such a tile before any compiler-visible VALU instruction has read it would see the accumulator before the MFMA has landed (round 2:
the last MFMA of a chain missing from the result, a 2e-5 error) - the sources keep a compiler-generated multiply in front of every
asm statement that consumes MFMA results, and this makes sure it stays there.
Usage: asm_mfma_waw_scan.py file.s        (exit status 1 if anything is found)
"""
import re, sys
from collections import deque
def asm_writes_into_recent_mfma_dest(txt, window=6, max_dist=48):
    """(function, line) of every instruction inside an inline-asm block whose destination VGPR lies inside the destination tile of one
    of the `window` most recently issued MFMAs."""
    return _scan(txt, window, max_dist, want_reads=False)


def asm_reads_of_unlanded_mfma_dest(txt, window=6, max_dist=48):
    """(function, line) of every instruction inside an inline-asm block that READS a VGPR inside the destination tile of one of the
    `window` most recently issued MFMAs which no compiler-visible VALU instruction has read yet."""
    return _scan(txt, window, max_dist, want_reads=True)


def _scan(txt, window, max_dist, want_reads):
    hits = []
    recent = deque(maxlen=window)
    in_asm = False
    fn = None
    pos = 0
    for ln in txt.splitlines():
        s = ln.strip()
        if re.match(r'(v_|s_|ds_|global_|buffer_|flat_)', s):
            pos += 1
        if re.match(r'^[A-Za-z_][\w$]*:', s):
            fn = s[:-1]; recent.clear(); in_asm = False
            continue
        if re.match(r'(s_branch|s_endpgm|s_setpc_b64)\b', s):
            recent.clear()                     # what follows in the file is not what follows in time
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True; continue
        if s.startswith(';;#ASMEND'):
            in_asm = False; continue
        m = re.match(r'v_mfma_\w+\s+v\[(\d+):(\d+)\]', s)
        if m:
            recent.append((int(m.group(1)), int(m.group(2)), pos)); continue
        if not in_asm and s.startswith('v_'):
            # a compiler-visible VALU read of an MFMA result carries the recogniser's wait states: that MFMA and (the matrix pipe
            # being in order) every older one have landed
            ops = s.split(None, 1)[1] if ' ' in s else ''
            srcs = ops.split(',')[1:]
            for o in srcs:
                for q in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', o):
                    lo = int(q.group(1) or q.group(3)); hi = int(q.group(2) or q.group(3))
                    for k in range(len(recent) - 1, -1, -1):
                        a, b, p = recent[k]
                        if lo <= b and hi >= a:
                            for _ in range(k + 1): recent.popleft()
                            break
            continue
        if in_asm and s.startswith('v_'):
            if want_reads:
                ops = s.split(None, 1)[1] if ' ' in s else ''
                found = False
                for o in ops.split(',')[1:]:
                    for q in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', o):
                        lo = int(q.group(1) or q.group(3)); hi = int(q.group(2) or q.group(3))
                        for a, b, p in recent:
                            if lo <= b and hi >= a and pos - p <= max_dist:
                                found = True
                if found:
                    hits.append((fn, s))
                continue
            d = re.match(r'v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))', s)
            if d:
                lo = int(d.group(1) or d.group(3)); hi = int(d.group(2) or d.group(3))
                for a, b, p in recent:
                    if lo <= b and hi >= a and pos - p <= max_dist:
                        hits.append((fn, s)); break
    return hits
if __name__ == '__main__':
    txt = open(sys.argv[1]).read()
    r = asm_reads_of_unlanded_mfma_dest(txt)
    print('%d asm reads of an MFMA destination no visible VALU instruction has read yet' % len(r))
    for x in r[:20]:
        print(x)
    h = asm_writes_into_recent_mfma_dest(txt)
    h = h + r
    print('%d asm-written registers inside the destination of a possibly running MFMA' % (len(h) - len(r)))
    for x in h[:20]:
        print(x)
    sys.exit(1 if h else 0)
