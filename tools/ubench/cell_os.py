#!/usr/bin/env python3
"""Fused cell with one vs two waves per (group, row tile) -- the output-channel split of round 6 (grouped_cell.hip, OS) -- at the
model's four block shapes and 4 ... 64 utterances: us per launch, outputs and statistics partials asserted bit-equal.

usage: python tools/ubench/cell_os.py [--batches 4 8 16 32 64] [--frames 1000] [--iters 20]
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
    ap.add_argument('--batches', type=int, nargs='+', default=[4, 8, 16, 32, 64])
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(0)
    t0 = a.frames
    for b in a.batches:
        for name, c, t in (('block_0', 600, t0), ('block_1', 800, t0), ('block_2', 1000, (t0 + 1) // 2), ('block_3', 1200, (t0 + 3) // 4)):
            groups, ld = 100, hip.round_up4(t)
            x = torch.zeros(b, c, ld, device=DEV)
            x[:, :, :t] = torch.randn(b, c, t, device=DEV) * 1.5 + 0.3
            stats = torch.empty(b, 2, ld, device=DEV)
            hip.channel_stats(x, stats, t, 1e-3)
            ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
            nodes = [(hip.pack_grouped_weights(torch.randn(c, c // groups, k, device=DEV) * 0.3, groups), torch.randn(c, device=DEV) * 0.2, k, d)
                     for k, d in ((7, 1), (5, 2), (7, 2))]
            res = {}
            for mode in ('0', '1'):
                os.environ['NBASR_CELL_OS'] = mode
                y, ws = torch.full_like(x, float('nan')), hip.grouped_stats_workspace(b, ld, groups, DEV)
                ws.zero_()
                ts = []
                for it in range(a.iters + 3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    hip.grouped_cell_fused(x, nodes, 0b111111, y, t, groups, ln, ws)
                    e1.record()
                    e1.synchronize()
                    if it >= 3:
                        ts.append(e0.elapsed_time(e1) * 1e3)
                res[mode] = (statistics.median(ts), y, ws)
            os.environ.pop('NBASR_CELL_OS')
            same = torch.equal(res['0'][1], res['1'][1]) and torch.equal(res['0'][2], res['1'][2])
            print(f'{name} b={b:2d} c={c} t={t}: one wave {res["0"][0]:7.1f} us   two waves {res["1"][0]:7.1f} us   bit-equal {same}', flush=True)
            assert same or c // groups in (6, 10)


if __name__ == '__main__':
    main()
