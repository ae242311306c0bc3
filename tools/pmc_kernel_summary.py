#!/usr/bin/env python3
"""Average of every collected counter per kernel name from a rocprofv3 counter_collection.csv.
Usage: pmc_kernel_summary.py <counter_collection.csv> [substring of kernel names to keep]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name).split('(')[0]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    key = (name, r.get('Grid_Size', ''))
    rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for (name, grid), cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print(f'{name[:70]} grid={grid} ({n} dispatches)')
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} {sum(v) / len(v):16.1f}')
