#!/usr/bin/env python3
"""Which pairs of torch streams execute kernels at the same time: the default stream, the first pool streams of normal and of high
priority, pairwise, with one-thread spin kernels (torch.cuda._sleep).  '.' = the two kernels overlapped, 'X' = one after the other.
usage: python tools/ubench/stream_pairs.py [--lp 8] [--hp 12]"""
import argparse
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument('--lp', type=int, default=8)
ap.add_argument('--hp', type=int, default=12)
a = ap.parse_args()
dev = torch.device('cuda', 0)
streams = [('null', torch.cuda.default_stream(dev))] + [(f'lp{i}', torch.cuda.Stream(device=dev)) for i in range(a.lp)] \
    + [(f'hp{i}', torch.cuda.Stream(device=dev, priority=-1)) for i in range(a.hp)]
CYC = 400_000
for _ in range(50):                     # clocks up
    torch.cuda._sleep(CYC)
torch.cuda.synchronize()


def run(ss, n=6):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in ss:
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


single = run([streams[1][1]])
print(f'one spin kernel: {1e6 * single:.0f} us')
names = [n for n, _ in streams]
print('      ' + ' '.join(f'{n:>4s}' for n in names))
for i, (ni, si) in enumerate(streams):
    row = []
    for j, (nj, sj) in enumerate(streams):
        if j <= i:
            row.append('    ')
            continue
        t = run([si, sj])
        row.append('   X' if t > 1.6 * single else '   .' if t < 1.3 * single else '   ?')
    print(f'{ni:>5s} ' + ' '.join(row), flush=True)
