#!/usr/bin/env python3
"""Print one steady-state training step of a rocprofv3 kernel trace as a timeline:
start offset, duration, queue, kernel (short name).  Usage: timeline.py trace.csv [step_index]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a step starts at the normal-distribution kernel that draws eps
    starts = [i for i, r in enumerate(rows) if 'normal_rows_kernel' in r['Kernel_Name'] or 'normal_and_transform' in r['Kernel_Name'] or 'distribution_elementwise' in r['Kernel_Name']]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
    lo, hi = starts[k], starts[k + 1]
    t0 = int(rows[lo]['Start_Timestamp'])
    queues = {}
    busy_end = 0
    for r in rows[lo:hi]:
        q = queues.setdefault(r['Queue_Id'], len(queues))
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        gap = s - busy_end if busy_end and s > busy_end else 0
        busy_end = max(busy_end, e)
        grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
        print(f"{s / 1e3:8.1f} {(e - s) / 1e3:7.1f}us q{q} {'  ' * q}{short(r['Kernel_Name'])}  [{grid} wg]" + (f"   <-- idle {gap / 1e3:.1f}us" if gap > 2000 else ''))
    print(f"step span {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us, {hi - lo} kernels")


if __name__ == '__main__':
    main()
