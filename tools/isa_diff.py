#!/usr/bin/env python3
"""Which kernels of two builds of the library are the same instruction stream?  Disassembles every code object of both (llvm-objdump), splits by
kernel symbol, strips addresses / encodings, and compares.  Usage: python tools/isa_diff.py OLD.so NEW.so [substring ...]   (kernels whose demangled
name contains every substring).  Used in round 5 to show that a feature compiled into the general per-env kernels left every default kernel alone."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import device_disassembly


def kernels(lib):
    out = {}
    for _, txt in device_disassembly(lib).items():
        for part in re.split(r'\n(?=[0-9a-f]{16} <)', txt):
            m = re.match(r'[0-9a-f]{16} <([^>]+)>:', part)
            if not m:
                continue
            body = []
            for ln in part.splitlines()[1:]:
                ln = re.sub(r'//.*$', '', ln).strip()
                ln = re.sub(r'<[^>]*>', '', ln)             # branch targets carry symbol+offset
                if ln:
                    body.append(ln)
            out[m.group(1)] = body
    return out


def demangle(names):
    try:
        return dict(zip(names, subprocess.run(['c++filt'] + names, check=True, capture_output=True, text=True).stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def main():
    old, new, want = kernels(sys.argv[1]), kernels(sys.argv[2]), sys.argv[3:]
    dem = demangle(sorted(set(old) | set(new)))
    same = diff = 0
    for k in sorted(new):
        name = re.sub(r'\(dpenv::.*$', '', re.sub(r'^void ', '', dem[k]))
        if not all(w in name for w in want):
            continue
        if k not in old:
            print('NEW       %s (%d instructions)' % (name[:110], len(new[k])))
        elif old[k] == new[k]:
            same += 1
            print('identical %s (%d instructions)' % (name[:110], len(new[k])))
        else:
            diff += 1
            print('DIFFERS   %s (%d -> %d instructions)' % (name[:110], len(old[k]), len(new[k])))
    print('%d identical, %d differ' % (same, diff))


if __name__ == '__main__':
    main()
