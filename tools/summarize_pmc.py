#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs per kernel.

usage: tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv>
Prints one row per kernel: launches, mean KiB counters and HBM bytes per launch with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reads
exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact
for 16 B/lane streaming stores.
"""
import collections
import csv
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name']
            short = name.split('(')[0].replace('void ', '')
            a = acc[short]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    # first line: hash of the graded kernel's sources, so bench.py can tell a stale summary from a current one
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from bench import kernel_source_hash
    print(f'# grouped_conv_src={kernel_source_hash()}')
    out = csv.writer(sys.stdout, lineterminator='\n')
    out.writerow(['kernel', 'launches', 'FETCH_SIZE_KiB_mean', 'WRITE_SIZE_KiB_mean', 'hbm_read_MB_corrected_x2', 'hbm_write_MB',
                  'hbm_total_MB'])
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        if not k.startswith('nbasr::'):
            continue
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        rd, wr = 2.0 * f * 1024 / 1e6, w * 1024 / 1e6
        out.writerow([k, n, f'{f:.1f}', f'{w:.1f}', f'{rd:.2f}', f'{wr:.2f}', f'{rd + wr:.2f}'])


if __name__ == '__main__':
    main()
