#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv to the rows that matter (this library's kernels and anything above
0.01 % of the total), same columns.  usage: tools/summarize_kernels.py <kernel_stats.csv>  > profiles/rNN_kernel_stats.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1], newline='')))
total = sum(float(r['TotalDurationNs']) for r in rows) or 1.0
w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0].keys()), lineterminator='\n')
w.writeheader()
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if 'nbasr::' in r['Name'] or float(r['TotalDurationNs']) / total >= 1e-4:
        w.writerow(r)
