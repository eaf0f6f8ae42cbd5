#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* counter_collection CSV per kernel: vector-ALU issue evidence for the grouped conv.

usage: tools/summarize_valu.py <counter_collection.csv>
Per kernel (mean over launches): VALU instructions per wave-cycle-quad, share of wave time spent issuing VALU, issue-stalled
(SQ_WAIT_INST_ANY) and parked (SQ_WAIT_ANY).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
(/opt/skills/guides/MI355X_MICROARCH.md, cycle-constants table), SQ_INSTS_* count instructions.
"""
import collections
import csv
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    with open(sys.argv[1], newline='') as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            if not name.startswith('nbasr::'):
                continue
            acc[name][row['Counter_Name']] += float(row['Counter_Value'])
            launches[name].add(row['Dispatch_Id'])
    out = csv.writer(sys.stdout, lineterminator='\n')
    cols = ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY',
            'SQ_WAIT_ANY']
    out.writerow(['kernel', 'launches'] + [c + '_per_launch' for c in cols] +
                 ['valu_active_frac_of_wave_cycles', 'any_active_frac', 'issue_stall_frac', 'parked_frac', 'salu_per_valu', 'smem_per_valu'])
    for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0.0)):
        a, n = acc[k], max(len(launches[k]), 1)
        wc = a.get('SQ_WAVE_CYCLES', 0.0) or float('nan')
        valu = a.get('SQ_INSTS_VALU', 0.0) or float('nan')
        out.writerow([k, n] + [f'{a.get(c, 0.0) / n:.4g}' for c in cols] +
                     [f'{a.get("SQ_ACTIVE_INST_VALU", 0.0) / wc:.3f}', f'{a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc:.3f}',
                      f'{a.get("SQ_WAIT_INST_ANY", 0.0) / wc:.3f}', f'{a.get("SQ_WAIT_ANY", 0.0) / wc:.3f}',
                      f'{a.get("SQ_INSTS_SALU", 0.0) / valu:.3f}', f'{a.get("SQ_INSTS_SMEM", 0.0) / valu:.3f}'])


if __name__ == '__main__':
    main()
