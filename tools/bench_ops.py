#!/usr/bin/env python3
"""Micro-benchmark of the individual HIP kernels at the model's shapes (B=64, T=1000 by default).

usage: python tools/bench_ops.py [dense|grouped|ln|lstm|head|all] [--batch 64] [--iters 20]
Prints per-op time (HIP events on the launch stream, median over iters), TFLOP/s or GB/s.
"""
import argparse
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip

DEV = 'cuda:0'


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', nargs='?', default='all')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    b, t = a.batch, a.frames
    t2, t3 = (t + 1) // 2, ((t + 1) // 2 + 1) // 2
    r4 = hip.round_up4
    torch.manual_seed(0)

    if a.what in ('dense', 'all'):
        for name, cin, cout, tin, s in (('conv_0', 80, 600, t, 1), ('conv_1', 600, 800, t, 1), ('conv_2', 800, 1000, t, 2), ('conv_3', 1000, 1200, t2, 2)):
            tout = (tin + s - 1) // s
            x = torch.randn(b, cin, r4(tin), device=DEV)
            w = torch.randn(cout, cin, 8, device=DEV) * 0.02
            bias = torch.randn(cout, device=DEV)
            y = torch.empty(b, cout, r4(tout), device=DEV)
            med, mn = timeit(lambda: hip.dense_conv1d_fused(x, tin, w, bias, (), y, s), a.iters)
            fl = 2.0 * b * tout * cout * cin * 8
            print(f'dense {name:7s} {cin:5d}->{cout:5d} T_out={tout:5d} s={s}: {med * 1e3:9.1f} us (min {mn * 1e3:9.1f})  {fl / med / 1e9:7.1f} TFLOP/s')
        for c, tt in ((600, t), (1200, t3)):
            x = torch.randn(b, c, r4(tt), device=DEV)
            w = torch.randn(c, c, 1, device=DEV) * 0.02
            bias = torch.randn(c, device=DEV)
            y = torch.empty_like(x)
            med, mn = timeit(lambda: hip.dense_conv1d_fused(x, tt, w, bias, (), y, 1), a.iters)
            fl = 2.0 * b * tt * c * c
            print(f'linear-op C={c:5d} T={tt:5d}: {med * 1e3:9.1f} us (min {mn * 1e3:9.1f})  {fl / med / 1e9:7.1f} TFLOP/s')

    if a.what in ('grouped', 'all'):
        for c, tt in ((600, t), (800, t), (1000, t2), (1200, t3)):
            for k, d in ((5, 1), (5, 2), (7, 1), (7, 2)):
                for ns in (0, 3):
                    x = torch.randn(b, c, r4(tt), device=DEV)
                    w = torch.randn(c, c // 100, k, device=DEV) * 0.1
                    bias = torch.randn(c, device=DEV)
                    sk = [torch.randn_like(x) for _ in range(ns)]
                    y = torch.empty_like(x)
                    med, mn = timeit(lambda: hip.grouped_conv1d_fused(x, w, bias, sk, y, tt, 100, k, d), a.iters)
                    by = 4.0 * (b * c * tt * (2 + ns) + c * (c // 100) * k + c)
                    fl = 2.0 * b * tt * c * (c // 100) * k
                    print(f'grouped C={c:5d} T={tt:5d} k={k} d={d} skips={ns}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})  {by / med / 1e6:7.0f} GB/s  {fl / med / 1e9:6.1f} TFLOP/s')

    if a.what in ('ln', 'all'):
        for c, tt in ((600, t), (800, t), (1000, t2), (1200, t3)):
            x = torch.randn(b, c, r4(tt), device=DEV)
            g, be = torch.rand(c, device=DEV), torch.randn(c, device=DEV)
            med, mn = timeit(lambda: hip.layernorm_channels(x, g, be, x, tt, 1e-3), a.iters)
            by = 8.0 * b * c * tt
            print(f'layernorm C={c:5d} T={tt:5d}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})  {by / med / 1e6:7.0f} GB/s (1R+1W algorithmic)')
            stats = torch.empty(b, 2, r4(tt), device=DEV)
            med, mn = timeit(lambda: hip.channel_stats(x, stats, tt, 1e-3), a.iters)
            print(f'channel_stats C={c:5d} T={tt:5d}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})  {by / 2 / med / 1e6:7.0f} GB/s (1R)')
            bound, image = torch.empty(b, device=DEV), hip.split_image(b, c, r4(tt), DEV)
            med, mn = timeit(lambda: hip.layernorm_split_image(x, g, be, stats, bound, image, tt, 1e-3), a.iters)
            print(f'layernorm_split_image (stats+bound, normalise+split) C={c:5d} T={tt:5d}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})')

    if a.what in ('lstm', 'all'):
        hid, cin = 500, 1200
        x = torch.randn(b, cin, r4(t3), device=DEV) * 0.5
        wih, whh = torch.randn(4 * hid, cin, device=DEV) * 0.03, torch.randn(4 * hid, hid, device=DEV) * 0.05
        bih, bhh = torch.randn(4 * hid, device=DEV) * 0.1, torch.randn(4 * hid, device=DEV) * 0.1
        gates = torch.empty(b * t3 * 4 * hid, device=DEV)
        cell = torch.empty(b * hid, device=DEV)
        h = torch.empty(b, t3, hid, device=DEV)
        med, mn = timeit(lambda: hip.lstm_forward(x, t3, wih, whh, bih, bhh, gates, cell, h), a.iters)
        print(f'lstm B={b} T={t3}: {med * 1e3:9.1f} us (min {mn * 1e3:9.1f})  = {med * 1e3 / t3:6.2f} us/step incl. input GEMM')

    if a.what in ('head', 'all'):
        h = torch.randn(b * t3, 500, device=DEV)
        w, bias = torch.randn(49, 500, device=DEV), torch.randn(49, device=DEV)
        out = torch.empty(b * t3, 49, device=DEV)
        med, mn = timeit(lambda: hip.linear_head(h, w, bias, out), a.iters)
        print(f'head rows={b * t3}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})')


if __name__ == '__main__':
    main()
