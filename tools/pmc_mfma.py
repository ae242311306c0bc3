#!/usr/bin/env python3
"""MFMA-pipe utilisation and wave-time split per kernel class from one rocprofv3 --pmc pass of bench.py
(counters SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).

    tools/pmc_mfma.py <counter_collection.csv> <out.json>

Under counter collection every dispatch runs alone, so these are stand-alone figures (no overlap between the
step's three streams).  GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs; MFMA utilisation =
sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)."""
import collections
import csv
import json
import sys

from pmc_traffic import CLASSES

SIMDS, XCDS = 1024, 8


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        for sub, cls in CLASSES:
            if sub in r['Kernel_Name']:
                acc[cls][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Dispatch_Id'] not in seen[cls]:
                    seen[cls].add(r['Dispatch_Id'])
                    acc[cls]['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                break
    out = {}
    for cls, c in acc.items():
        gui = c['GRBM_GUI_ACTIVE'] / XCDS
        wave = max(c['SQ_WAVE_CYCLES'], 1.0)
        out[cls] = {'launches': len(seen[cls]), 'avg_us_alone': c['ns'] / len(seen[cls]) / 1e3,
                    'mfma_util_percent': 100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * SIMDS) if gui else None,
                    'wave_time_waiting_percent': 100.0 * c['SQ_WAIT_ANY'] / wave,
                    'wave_time_issue_stalled_percent': 100.0 * c['SQ_WAIT_INST_ANY'] / wave,
                    'wave_time_issuing_percent': 100.0 * c['SQ_ACTIVE_INST_ANY'] / wave}
    json.dump({'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY '
                         'GRBM_GUI_ACTIVE -- python3 bench.py --steps 20 --warmup 5 (dispatches run alone under counter collection)',
               'classes': out}, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    for cls, v in sorted(out.items()):
        print(f"{cls:20s} {v['launches']:5d} launches  {v['avg_us_alone']:7.1f} us alone  MFMA util {v['mfma_util_percent'] or 0:5.1f} %  "
              f"waiting {v['wave_time_waiting_percent']:5.1f} %")


if __name__ == '__main__':
    main()
