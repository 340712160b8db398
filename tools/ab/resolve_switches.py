#!/usr/bin/env python3
"""Partial preprocessor used ONCE in round 6 to retire the A/B build switches of rejected variants from the product sources
(VERDICT r05 item 7): every `#if` / `#ifdef` / `#ifndef` whose condition involves only the macros below is resolved to the shipped value, the
default-definition blocks go, and uses of the macro in code become the literal.  Everything else passes through untouched.  The reverse patch
(tools/ab/r06_build_switches.patch) re-introduces the switches on the commit it was taken from.
Usage: python tools/ab/resolve_switches.py FILE ...   (rewrites in place, prints what it dropped)"""
import re
import sys

VALUE = {'DPENV_STEP_HOIST_LOADS': '1', 'DPENV_STEP_STATE_STORES_FIRST': '1', 'DPENV_STEP_PRELOAD_ARGS': '0', 'DPENV_WS_POLL_SLEEP': '2',
         'DPENV_WS_STAGE_ACTOR': '0', 'DPENV_WS_X_MNOISE': '0', 'DPENV_WS_ECRITIC_MNOISE': '1', 'DPENV_WS_F16_G2_MNOISE': '0',
         'DPENV_WS_M_PRIO': '3', 'DPENV_WS_C_PRIO': '1', 'DPENV_WS_PREDRAW': '1', 'DPENV_JOINT_EVAL': '1'}
UNDEF = {'DPENV_WS_DEBUG_NOMFMA', 'DPENV_WS_SWAP_ROLES', 'DPENV_WS_NO_SETPRIO', 'DPENV_WS_M_PRIO_CRITIC', 'DPENV_WS_X_CARRY_FRAGS',
         'DPENV_WS_E_PRIO', 'DPENV_WS_DYN_PRIO'}
KNOWN = set(VALUE) | UNDEF
DIR = re.compile(r'^\s*#\s*(if|ifdef|ifndef|elif|else|endif|define)\b(.*)$')


def resolvable(cond):
    names = set(re.findall(r'[A-Za-z_]\w*', cond)) - {'defined'}
    return bool(names) and names <= KNOWN


def evaluate(cond):
    cond = re.sub(r'//.*$', '', cond)
    cond = re.sub(r'defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)', lambda m: '1' if (m.group(1) or m.group(2)) in VALUE else '0', cond)
    cond = re.sub(r'[A-Za-z_]\w*', lambda m: VALUE.get(m.group(0), '0'), cond)
    cond = cond.replace('&&', ' and ').replace('||', ' or ')
    cond = re.sub(r'!(?!=)', ' not ', cond)
    return bool(eval(cond))


def run(path):
    out, stack = [], []
    for ln in open(path).read().split('\n'):
        m = DIR.match(ln)
        live = all(f['active'] for f in stack if f['res'])
        if m:
            d, rest = m.group(1), m.group(2).strip()
            if d in ('if', 'ifdef', 'ifndef'):
                cond = rest if d == 'if' else ('defined(%s)' % rest.split()[0] if d == 'ifdef' else '!defined(%s)' % rest.split()[0])
                if resolvable(re.sub(r'//.*$', '', cond)):
                    v = evaluate(cond)
                    stack.append({'res': True, 'active': v, 'taken': v})
                    print('%s: resolved `%s` -> %s' % (path, ln.strip(), v))
                    continue
                stack.append({'res': False})
            elif d in ('elif', 'else'):
                f = stack[-1]
                if f['res']:
                    f['active'] = (not f['taken']) and (True if d == 'else' else evaluate(rest))
                    f['taken'] = f['taken'] or f['active']
                    continue
            elif d == 'endif':
                f = stack.pop()
                if f['res']:
                    continue
            elif d == 'define' and rest.split()[0].split('(')[0] in KNOWN:
                if live:
                    print('%s: dropped `%s`' % (path, ln.strip()))
                continue
            if live:
                out.append(ln)
            continue
        if live:
            new = re.sub(r'\bDPENV_[A-Z_0-9]+\b', lambda mm: VALUE.get(mm.group(0), mm.group(0)), ln)
            for u in UNDEF:
                assert not re.search(r'\b%s\b' % u, re.sub(r'//.*$', '', new)), (path, ln)
            out.append(new)
    assert not stack, path
    open(path, 'w').write('\n'.join(out))


for p in sys.argv[1:]:
    run(p)
