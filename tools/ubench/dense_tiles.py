#!/usr/bin/env python3
"""Time the image-path dense convolution (fp16 split) at every (row tile, frame tile) for the four downsample convs at one batch size:
the table behind executor._dense_tile.  Outputs of every tiling are asserted bit-equal (the K order of an output does not depend on it).

usage: python tools/ubench/dense_tiles.py [--batch 8] [--frames 1000] [--iters 15]
"""
import argparse
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip

DEV = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=15)
    a = ap.parse_args()
    b, t = a.batch, a.frames
    t2 = (t + 1) // 2
    torch.manual_seed(0)
    for name, cin, cout, tin, s in (('conv_0', 80, 600, t, 1), ('conv_1', 600, 800, t, 1), ('conv_2', 800, 1000, t, 2), ('conv_3', 1000, 1200, t2, 2)):
        ld = hip.round_up4(tin)
        x = torch.zeros(b, cin, ld, device=DEV)
        x[:, :, :tin] = torch.randn(b, cin, tin, device=DEV) * 2.0 + 0.3
        g, be = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.2
        stats, bound = torch.empty(b, 2, ld, device=DEV), torch.empty(b, device=DEV)
        image = hip.split_image(b, cin, ld, DEV)
        hip.layernorm_split_image(x, g, be, stats, bound, image, tin, 1e-3)
        w = torch.randn(cout, cin, 8, device=DEV) * (2.0 / (cin * 8)) ** 0.5
        bias = torch.randn(cout, device=DEV) * 0.1
        tout = (tin + s - 1) // s
        ld_out = hip.round_up4(tout)
        combos = [(r, ft) for r in (64, 96, 128, 160) for ft in (256, 128)]
        packed = {r: hip.pack_dense_weights(w, s, 'f16x2', row_tile=r) for r in (64, 96, 128, 160)}
        part = torch.empty(hip.dense_stats_part_floats(b, cout, ld_out), device=DEV)
        ref = None
        times = {c: [] for c in combos}

        def run(r, ft, y):
            hip.dense_conv1d_fused_packed_f16_img(image, bound, b, cin, tin, ld, packed[r], cout, 8, bias, y, s, row_tile=r, stats_part=part, frame_tile=ft)

        for r, ft in combos:
            y = torch.full((b, cout, ld_out), float('nan'), device=DEV)
            run(r, ft, y)
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            assert torch.equal(y, ref), (name, r, ft)
        y = torch.empty_like(ref)
        for it in range(a.iters + 2):
            for c in combos:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(c[0], c[1], y)
                e1.record()
                e1.synchronize()
                if it >= 2:
                    times[c].append(e0.elapsed_time(e1) * 1e3)
        best = min(combos, key=lambda c: statistics.median(times[c]))
        for r, ft in combos:
            wgs = -(-cout // r) * -(-ld_out // ft) * b
            print(f'{name} b={b} rows={r:3d} frames={ft:3d} workgroups={wgs:5d}: {statistics.median(times[(r, ft)]):8.1f} us'
                  + ('   <-- best' if (r, ft) == best else ''), flush=True)


if __name__ == '__main__':
    main()
