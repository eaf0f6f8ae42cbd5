#!/usr/bin/env python3
"""How many kernels from different streams execute at once: S streams x N spin kernels of one thread each (torch.cuda._sleep).
If S streams take as long as one, the S kernels overlapped.   usage: python tools/ubench/stream_concurrency.py"""
import time

import torch

dev = torch.device('cuda', 0)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
N, CYC = 100, 200_000
for prio in (0, -1):
    for S in (1, 2, 3, 4, 6, 8):
        streams = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(S)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            for s in streams:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'priority {prio:2d}: {S} streams x {N} spin kernels: {1e3 * dt:7.2f} ms  ({1e6 * dt / N:6.1f} us per round)', flush=True)
