#!/usr/bin/env python3
"""Static scan of gfx950 ISA (hipcc -S output) for short producer -> consumer distances.

For every VALU instruction of interest (default: packed-fp32 v_pk_*), list each source VGPR/SGPR whose most recent
writer in straight-line program order sits within `--window` issued instructions, with the writer's opcode.  Used to
look for missing wait states (trans -> consumer, v_readlane -> SGPR consumer, LDS return -> consumer without a wait).
Usage: isa_hazard_scan.py file.s [--kernel SUBSTR] [--pattern REGEX] [--window N]
"""
import re, sys, argparse

TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')

def regs_of(tok):
    """expand an operand token to a list of register names"""
    tok = tok.strip()
    tok = re.sub(r'^[-|]+|[|]+$', '', tok)
    m = re.match(r'^([vsa])\[(\d+):(\d+)\]$', tok)
    if m:
        return ['%s%d' % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r'^([vsa])(\d+)$', tok)
    if m:
        return [tok]
    if tok in ('vcc', 'exec'):
        return [tok]
    return []

def parse(line):
    line = line.split(';')[0].strip()
    if not line or line.endswith(':') or line.startswith('.'):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = []
    if len(parts) > 1:
        rest = re.sub(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[[^\]]*\]', '', parts[1])
        rest = re.sub(r'\b(offset\d?|offset):\d+', '', rest)
        # split on commas not inside brackets
        depth = 0; cur = ''
        for ch in rest:
            if ch == '[': depth += 1
            if ch == ']': depth -= 1
            if ch == ',' and depth == 0:
                ops.append(cur); cur = ''
            else:
                cur += ch
        ops.append(cur)
        ops = [o.strip().split()[0] if o.strip() else '' for o in ops]
    return op, ops

def n_dst(op):
    if op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_sleep', 's_setprio', 's_barrier', 's_endpgm',
                      'global_store', 'ds_write', 'buffer_store', 's_cmp', 's_bitcmp')):
        return 0
    if op.startswith('v_permlane32_swap') or op.startswith('v_swap'):
        return 2
    if op.startswith(('v_cmp',)) and op.endswith('_e32'):
        return 0      # writes vcc implicitly
    return 1

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('file'); ap.add_argument('--kernel', default=None)
    ap.add_argument('--pattern', default=r'^v_pk_'); ap.add_argument('--window', type=int, default=3)
    ap.add_argument('--from-line', type=int, default=0); ap.add_argument('--to-line', type=int, default=10 ** 9)
    a = ap.parse_args()
    pat = re.compile(a.pattern)
    lines = open(a.file).read().split('\n')
    inside = a.kernel is None
    hist = []          # (lineno, op, dsts)
    found = 0
    for ln, raw in enumerate(lines, 1):
        if a.kernel is not None:
            if raw.startswith('_Z') and raw.rstrip().split(':')[0].find(a.kernel) >= 0 and ':' in raw:
                inside = True; hist = []
            elif raw.startswith('.Lfunc_end'):
                inside = False
        if not inside or ln < a.from_line or ln > a.to_line:
            continue
        if re.match(r'^\.LBB', raw):
            hist.append((ln, 'LABEL', []))          # control-flow join: distances across it are lower bounds only
            continue
        p = parse(raw)
        if p is None:
            continue
        op, ops = p
        nd = n_dst(op)
        dsts = [r for t in ops[:nd] for r in regs_of(t)]
        if op.startswith('v_cmp') and op.endswith('_e32'):
            dsts = ['vcc']
        srcs = [r for t in ops[nd:] for r in regs_of(t)]
        if op.startswith(('v_fmac', 'v_pk_fmac', 'v_mac')):
            srcs += dsts
        if pat.search(op):
            # wait states = number of issued instructions between (s_nop N counts N+1)
            for s in sorted(set(srcs)):
                ws = 0
                for (pl, pop, pd) in reversed(hist):
                    if pop == 'LABEL':
                        break
                    if s in pd:
                        if ws <= a.window:
                            tag = 'TRANS' if pop.startswith(TRANS) else ('LANE' if 'lane' in pop else ('LDS' if pop.startswith('ds_') else ('MEM' if pop.startswith(('global_', 'buffer_', 's_load')) else '')))
                            print('%6d %-34s src %-5s <- %-28s (line %d) wait_states=%d %s' % (ln, op, s, pop, pl, ws, tag))
                            found += 1
                        break
                    m = re.match(r'^s_nop', pop)
                    ws += 1
                    if m:
                        pass
        # s_nop N occupies N+1 wait states: record as N+1 pseudo-entries
        if op == 's_nop':
            k = int(ops[0]) if ops and ops[0].isdigit() else 0
            for _ in range(k + 1):
                hist.append((ln, 's_nop', []))
        else:
            hist.append((ln, op, dsts))
    print('entries:', found)

if __name__ == '__main__':
    main()
