#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY run.

usage: tools/summarize_mfma.py <counter_collection.csv> [--all]      (--all: also kernels that issue no MFMA; their clock matters too)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  clock_GHz = kernel cycles / duration;  parked = 4 * SQ_WAIT_ANY / (4 * SQ_WAVE_CYCLES)   (quad-cycle counters)
"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
ALL = '--all' in sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not name.startswith('nbasr::'):
        continue
    key = (name, r['Grid_Size'])
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    acc[key]['dur_ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = csv.writer(sys.stdout, lineterminator='\n')
out.writerow(['kernel', 'grid_threads', 'launches', 'avg_us', 'clock_GHz', 'mfma_busy_frac', 'waves_parked_frac'])
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]['dur_ns'])):
    mean = {k: sum(x) / len(x) for k, x in v.items()}
    if mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) == 0 and not ALL:
        continue
    cycles = mean['GRBM_GUI_ACTIVE'] / 8.0
    out.writerow([name, grid, len(v['GRBM_GUI_ACTIVE']), f"{mean['dur_ns'] / 1e3:.1f}", f"{cycles / mean['dur_ns']:.2f}",
                  f"{mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cycles):.3f}",
                  f"{mean['SQ_WAIT_ANY'] / mean['SQ_WAVE_CYCLES']:.3f}" if mean.get('SQ_WAVE_CYCLES') else ''])
