#!/usr/bin/env python3
"""The bf16 matrix-core cell alone at BASELINE config 4's per-GPU shapes (32 x 1600, conv7 / conv7d2 / conv5d2, all nine skips, a
pending LayerNorm on load): a chain of launches through buffers larger than the last-level cache.

usage: python tools/ubench/bench_cell_mfma.py [--batch 32] [--frames 1600] [--mask 63] [--iters 30]"""
import argparse
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from nb_asr_amd import hip

DEV = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=1600)
    ap.add_argument('--mask', type=int, default=63)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--kds', default='7,1;7,2;5,2')
    ap.add_argument('--sweep', action='store_true', help='time every tiling (NBASR_CELLM_TILING) per block shape')
    ap.add_argument('--lib', default=None, help='another build of libnbasr_hip.so to time (A/B)')
    a = ap.parse_args()
    if a.lib:
        hip._lib = hip.load_library(a.lib)
    kds = [tuple(map(int, s.split(','))) for s in a.kds.split(';')]
    b, t = a.batch, a.frames
    t2, t3 = (t + 1) // 2, ((t + 1) // 2 + 1) // 2
    torch.manual_seed(0)
    out = []
    for blk, (c, tt) in enumerate(((600, t), (800, t), (1000, t2), (1200, t3))):
        ld = hip.row_pitch(tt, torch.bfloat16)
        nbuf = max(3, int(600e6 // (b * c * ld * 2)) + 1)
        bufs = [(torch.randn(b, c, ld, device=DEV) * 1.5).to(torch.bfloat16) for _ in range(nbuf)]
        for x in bufs:
            x[:, :, tt:] = 0
        nodes = [(hip.grouped_cell_mfma_pack(torch.randn(c, c // 100, k, device=DEV) * 0.1, 100), torch.randn(c, device=DEV) * 0.2, k, d) for k, d in kds]
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(bufs[0], stats, tt, 1e-3)
        ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
        gpw = hip.grouped_cell_mfma_fits(c, ld, 100)

        def step(i):
            hip.grouped_cell_mfma(bufs[i % nbuf], nodes, a.mask, bufs[(i + 1) % nbuf], tt, 100, ln)
        if a.sweep:
            import os
            for nbt in (8, 10, 14, 16):
                for g in (1, 2, 4):
                    os.environ['NBASR_CELLM_TILING'] = f'{nbt},{g}'
                    try:
                        for i in range(3):
                            step(i)
                    except hip.HipError:
                        continue
                    torch.cuda.synchronize()
                    ts = []
                    for i in range(a.iters):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); step(i); e1.record(); e1.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3)
                    print(f'   block {blk} nbt={nbt} gpw={g}: {statistics.median(ts):7.1f} us', flush=True)
            del os.environ['NBASR_CELLM_TILING']
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        ts = []
        for i in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step(i)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        fl = sum(2.0 * b * tt * c * (c // 100) * k for k, _ in kds)
        med = statistics.median(ts)
        out.append(med)
        print(f'block {blk} C={c} T={tt} gpw={gpw}: {med:7.1f} us (min {min(ts):7.1f})  {fl / med / 1e6:6.1f} TFLOP/s  {2 * b * c * tt * 2 / med / 1e3:6.0f} GB/s', flush=True)
    print('sum x (3,4,5,6):', round(3 * out[0] + 4 * out[1] + 5 * out[2] + 6 * out[3], 1), 'us per forward')


if __name__ == '__main__':
    main()
