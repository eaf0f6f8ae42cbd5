#!/usr/bin/env python3
"""Node kernel timing, independent buffers against a producer -> consumer chain (every launch reads what the previous one
wrote, as inside the model), per block of the benchmark configuration.

    python tools/ubench/gc_chain.py
"""
import json
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip  # noqa: E402

dev = torch.device('cuda', 0)
B, t = 64, 1000
for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
    t = (t + stride - 1) // stride
    ld = (t + 3) & ~3
    w = torch.randn(c, c // 100, 5, device=dev) * 0.2
    bias = torch.randn(c, device=dev) * 0.1
    row = {'block': blk, 'C': c, 'T': t}
    for nbuf in (2, 3, 4, 8):
        bufs = [torch.randn(B, c, ld, device=dev) * 0.5 for _ in range(nbuf)]
        for b_ in bufs:
            b_[:, :, t:] = 0
        big = torch.empty(64 * 1024 * 1024, device=dev)            # 256 MB: flushed through the last-level cache between modes

        def chain(i):
            hip.grouped_conv1d_node(bufs[i % nbuf], w, bias, [], bufs[(i + 1) % nbuf], t, 100, 5, 1, None, False, False, None, 0)

        def indep(i):
            k = (2 * i) % nbuf
            hip.grouped_conv1d_node(bufs[k], w, bias, [], bufs[(k + 1) % nbuf], t, 100, 5, 1, None, False, False, None, 0)

        for name, fn in (('chain', chain), ('indep', indep)):
            if name == 'indep' and nbuf < 4:
                continue
            big.zero_()
            for i in range(6):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 60
            e0.record()
            for i in range(n):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            row[f'{name}_{nbuf}buf_us'] = round(e0.elapsed_time(e1) * 1e3 / n, 1)
        del bufs, big
    print(json.dumps(row), flush=True)
