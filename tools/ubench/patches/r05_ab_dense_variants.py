#!/usr/bin/env python3
"""A/B of the image-path dense convolution's pipeline variants (NBASR_DENSE_VARIANT, experiment builds only) at the four
downsample convs of the benchmark shape: time per launch (HIP events, alternating launches, median) and bit-equality with
variant 0.

usage: python tools/ubench/ab_dense_variants.py [--batch 64] [--frames 1000] [--variants 0,1,2] [--iters 15]
"""
import argparse
import os
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip

DEV = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--variants', default='0,1,2,3,4,5')
    ap.add_argument('--iters', type=int, default=15)
    ap.add_argument('--rows', default='')
    ap.add_argument('--zeros', action='store_true', help='all-zero operands: the same instruction stream without the data-dependent power draw')
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(',')]
    b, t = a.batch, a.frames
    t2 = (t + 1) // 2
    shapes = [('conv_0', 80, 600, t, 1, 128), ('conv_1', 600, 800, t, 1, 160), ('conv_2', 800, 1000, t, 2, 128), ('conv_3', 1000, 1200, t2, 2, 160)]
    if a.rows:
        rows = [int(r) for r in a.rows.split(',')]
        shapes = [(n, ci, co, tt, s, r) for (n, ci, co, tt, s, _), r in zip(shapes, rows)]
    torch.manual_seed(0)
    for name, cin, cout, tin, s, rows in shapes:
        ld = hip.round_up4(tin)
        x = torch.zeros(b, cin, ld, device=DEV)
        x[:, :, :tin] = torch.randn(b, cin, tin, device=DEV) * 2.0 + 0.3
        g, be = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.2
        stats, bound = torch.empty(b, 2, ld, device=DEV), torch.empty(b, device=DEV)
        image = hip.split_image(b, cin, ld, DEV)
        hip.layernorm_split_image(x, g, be, stats, bound, image, tin, 1e-3)
        w = torch.randn(cout, cin, 8, device=DEV) * (2.0 / (cin * 8)) ** 0.5
        if a.zeros:
            image.zero_()
            w.zero_()
        bias = torch.randn(cout, device=DEV) * 0.1
        tout = (tin + s - 1) // s
        packed = hip.pack_dense_weights(w, s, 'f16x2', row_tile=rows)
        outs, times = {}, {v: [] for v in variants}
        ok = {}
        for v in variants:
            os.environ['NBASR_DENSE_VARIANT'] = str(v)
            y = torch.full((b, cout, hip.round_up4(tout)), float('nan'), device=DEV)
            try:
                hip.dense_conv1d_fused_packed_f16_img(image, bound, b, cin, tin, ld, packed, cout, 8, bias, y, s, row_tile=rows)
                torch.cuda.synchronize()
                ok[v] = True
            except hip.HipError as e:
                ok[v] = False
                print(f'  {name} variant {v}: {e}')
            outs[v] = y
        y = torch.empty_like(outs[variants[0]])
        for it in range(a.iters + 2):
            for v in variants:
                if not ok[v]:
                    continue
                os.environ['NBASR_DENSE_VARIANT'] = str(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                hip.dense_conv1d_fused_packed_f16_img(image, bound, b, cin, tin, ld, packed, cout, 8, bias, y, s, row_tile=rows)
                e1.record()
                e1.synchronize()
                if it >= 2:
                    times[v].append(e0.elapsed_time(e1) * 1e3)
        fl = 2.0 * b * tout * cout * cin * 8
        for v in variants:
            if not ok[v]:
                continue
            med = statistics.median(times[v])
            same = bool(torch.equal(outs[v], outs[variants[0]]))
            print(f'{name} rows={rows} s={s} variant {v}: {med:8.1f} us (min {min(times[v]):8.1f})  {3 * fl / med / 1e6:7.1f} TF issued  bit-equal to v{variants[0]}: {same}', flush=True)
    os.environ.pop('NBASR_DENSE_VARIANT', None)


if __name__ == '__main__':
    main()
