#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv under a directory): one line per kernel x counter.
    python tools/pmc_by_kernel.py <dir> [substring-of-kernel-name]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if want and want not in k:
                continue
            key = (k[:110], r.get('Grid_Size', ''), r['Counter_Name'])
            a = acc[key]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    for (k, g, c), (s, n) in sorted(acc.items()):
        print('%-110s grid %-9s %-28s %14.1f  (%d dispatches)' % (k, g, c, s / n, n))


if __name__ == '__main__':
    main()
