#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> how busy each hardware queue was and how many kernels ran at once (the W-chains-in-flight experiments).
usage: python tools/ubench/trace_concurrency.py kernel_trace.csv [t_from_fraction t_to_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.6, 0.95)
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r.get('Stream_Id', '?'), r['Kernel_Name']) for r in rows))
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
win = [e for e in ev if e[0] >= a and e[1] <= b]
span = (b - a) / 1e6
print(f'{len(ev)} kernels over {(t_hi - t_lo) / 1e6:.1f} ms; window {span:.2f} ms with {len(win)} kernels')
per_q = defaultdict(lambda: [0, 0.0])
for s, e, q, st, n in win:
    per_q[(q, st)][0] += 1
    per_q[(q, st)][1] += (e - s) / 1e6
for k, (n, busy) in sorted(per_q.items()):
    print(f'  queue {k[0]} stream {k[1]}: {n:6d} kernels, busy {busy:8.2f} ms = {busy / span:.2f} of the window')
pts = sorted([(s, 1) for s, e, *_ in win] + [(e, -1) for s, e, *_ in win])
cur, last, hist = 0, a, defaultdict(float)
for t, d in pts:
    hist[cur] += t - last
    cur, last = cur + d, t
tot = sum(hist.values())
print('  kernels running at once: ' + '  '.join(f'{k}: {v / tot:.2f}' for k, v in sorted(hist.items())))
