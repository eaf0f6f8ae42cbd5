#!/usr/bin/env python3
"""The fused cell alone at the benchmark's four block shapes: a chain of launches through buffers larger than the last-level cache,
pending LayerNorm on load and the statistics by-product as in the model.

usage: python tools/bench_cell.py [--batch 64] [--frames 1000] [--kd 5,1] [--mask 0] [--iters 30]"""
import argparse
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip

DEV = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--kd', default='5,1')
    ap.add_argument('--mask', type=int, default=0)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    k, d = map(int, a.kd.split(','))
    b, t = a.batch, a.frames
    t2, t3 = (t + 1) // 2, ((t + 1) // 2 + 1) // 2
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    torch.manual_seed(0)
    out = []
    for blk, (c, tt) in enumerate(((600, t), (800, t), (1000, t2), (1200, t3))):
        ld = hip.row_pitch(tt, dt)
        nbuf = max(3, int(600e6 // (b * c * ld * 4)) + 1)
        bufs = [(torch.randn(b, c, ld, device=DEV) * 1.5).to(dt) for _ in range(nbuf)]
        for x in bufs:
            x[:, :, tt:] = 0
        nodes = [(hip.pack_grouped_weights(torch.randn(c, c // 100, k, device=DEV) * 0.3, 100), torch.randn(c, device=DEV) * 0.2, k, d) for _ in range(3)]
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(bufs[0], stats, tt, 1e-3)
        ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
        ws = hip.grouped_stats_workspace(b, ld, 100, DEV)
        gpp = hip.grouped_cell_fits(c, ld, 100)

        def step(i):
            hip.grouped_cell_fused(bufs[i % nbuf], nodes, a.mask, bufs[(i + 1) % nbuf], tt, 100, ln, ws)
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
        fl = 3 * 2.0 * b * tt * c * (c // 100) * k
        med = statistics.median(ts)
        out.append(med)
        print(f'block {blk} C={c} T={tt} gpp={gpp}: {med:7.1f} us (min {min(ts):7.1f})  {fl / med / 1e6:6.1f} TFLOP/s  {2 * b * c * tt * bufs[0].element_size() / med / 1e3:6.0f} GB/s', flush=True)
    print('sum x (3,4,5,6):', round(3 * out[0] + 4 * out[1] + 5 * out[2] + 6 * out[3], 1), 'us per forward')


if __name__ == '__main__':
    main()
