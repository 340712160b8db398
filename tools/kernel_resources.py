#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a built libdpenv.so (or any .o / .so with bundled gfx9xx code objects), read from the
code objects' own metadata (llvm-readelf --notes: .vgpr_count, .agpr_count, .sgpr_count, .private_segment_fixed_size,
.group_segment_fixed_size, .max_flat_workgroup_size) - what the hardware is told, independent of any profiler's rounding.

    python tools/kernel_resources.py ml4ca_amd/lib/libdpenv.so [substring ...]      one line per kernel whose name has every substring
    python tools/kernel_resources.py --json out.json lib.so                         the same as {demangled name: record}
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import OBJDUMP          # noqa: E402

BIN = os.path.dirname(OBJDUMP)


def resources(lib):
    tmp = tempfile.mkdtemp(prefix='dpenv_res_')
    out = {}
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([OBJDUMP, '--offloading', work], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if 'amdgcn' not in f:
                continue
            txt = subprocess.run([os.path.join(BIN, 'llvm-readelf'), '--notes', os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
                blk = '.agpr_count:' + blk
                g = lambda k: (re.search(r'\.%s:\s*(\S+)' % k, blk) or [None, None])[1]
                name = g('name')
                if not name:
                    continue
                out[name] = {'vgpr': int(g('vgpr_count') or 0), 'agpr': int(g('agpr_count') or 0), 'sgpr': int(g('sgpr_count') or 0),
                             'scratch_bytes_per_lane': int(g('private_segment_fixed_size') or 0), 'lds_static_bytes': int(g('group_segment_fixed_size') or 0),
                             'max_workgroup': int(g('max_flat_workgroup_size') or 0)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = list(out)
    try:
        filt = os.path.join(BIN, 'llvm-cxxfilt')
        dem = subprocess.run([filt if os.path.exists(filt) else 'c++filt'] + names, check=True, capture_output=True, text=True).stdout.splitlines()
        out = {re.sub(r'\(dpenv::.*$', '', re.sub(r'^void ', '', d)): out[n] for d, n in zip(dem, names)}
    except Exception:
        pass
    return out


def main():
    args = sys.argv[1:]
    js = None
    if args and args[0] == '--json':
        js, args = args[1], args[2:]
    res = resources(args[0])
    want = args[1:]
    if js:
        json.dump(res, open(js, 'w'), indent=1, sort_keys=True)
    for name in sorted(res):
        if all(w in name for w in want):
            r = res[name]
            print('%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d B/lane  lds %6d B  wg <= %d' % (name[:100], r['vgpr'], r['agpr'], r['sgpr'], r['scratch_bytes_per_lane'],
                                                                                          r['lds_static_bytes'], r['max_workgroup']))


if __name__ == '__main__':
    main()
