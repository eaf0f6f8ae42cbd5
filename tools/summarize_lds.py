#!/usr/bin/env python3
"""Per-kernel means of the LDS counters of a rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES run.

usage: tools/summarize_lds.py <counter_collection.csv>
  bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  (extra LDS-array cycles of conflicts over all LDS-array cycles,
  /opt/skills/guides/MI355X_MICROARCH.md, LDS section)
"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not name.startswith('nbasr::'):
        continue
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    acc[name]['dur_ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
names = ['SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_WAVE_CYCLES']
out = csv.writer(sys.stdout, lineterminator='\n')
out.writerow(['kernel', 'launches', 'avg_us'] + [n + '_per_launch' for n in names] + ['bank_conflict_frac_of_lds_active'])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]['dur_ns'])):
    mean = {n: (sum(x) / len(x)) for n, x in v.items()}
    if mean.get('SQ_LDS_IDX_ACTIVE', 0) == 0:
        continue
    n_l = max(len(x) for n, x in v.items() if n != 'dur_ns')
    out.writerow([k, n_l, f"{mean['dur_ns'] / 1e3:.1f}"] + [f"{mean.get(n, 0):.4g}" for n in names] +
                 [f"{mean.get('SQ_LDS_BANK_CONFLICT', 0) / mean['SQ_LDS_IDX_ACTIVE']:.3f}"])
