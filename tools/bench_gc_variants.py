#!/usr/bin/env python3
"""A/B of the grouped-conv node kernel's variants (storage type x frames per lane x weight layout) at the model's shapes.

    python tools/bench_gc_variants.py [--batch 64] [--frames 1000] [--kernel 5] [--dilation 1]

Interleaved rounds in ONE process (same device, same clocks): for every block shape (C, T_b) and node flavour (plain /
LayerNorm on load / statistics epilogue / three skips) time each variant `--rounds` times, report the median and TB/s of
algorithmic bytes.  Results are bit-identical across variants (tests/test_bf16_ops_gpu.py); this is speed only.
"""
import argparse
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--kernel', type=int, default=5)
    ap.add_argument('--dilation', type=int, default=1)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--dtypes', default='f32,bf16')
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    out = []
    t = a.frames
    for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
        t = (t + stride - 1) // stride
        for dname in a.dtypes.split(','):
            dtype = torch.float32 if dname == 'f32' else torch.bfloat16
            elem = 4 if dname == 'f32' else 2
            ld = hip.row_pitch(t, torch.bfloat16)                 # a pitch every variant accepts
            mk = lambda: (torch.randn(a.batch, c, ld, device=dev) * 0.5).to(dtype)       # noqa: E731
            x, y, s0, s1, s2 = mk(), mk(), mk(), mk(), mk()
            for tns in (x, s0, s1, s2):
                tns[:, :, t:] = 0
            w = torch.randn(c, c // 100, a.kernel, device=dev) * 0.2
            wp = hip.pack_grouped_weights(w, 100)
            bias = torch.randn(c, device=dev) * 0.1
            stats = torch.empty(a.batch, 2, ld, device=dev)
            hip.channel_stats(x, stats, t, 1e-3)
            ln = (stats, torch.ones(c, device=dev), torch.zeros(c, device=dev))
            ws = hip.grouped_stats_workspace(a.batch, ld, 100, dev)
            flavours = {'plain': dict(skips=[], ln=None, on_x=False, ws=None),
                        'lnx': dict(skips=[], ln=ln, on_x=True, ws=None),
                        'stats': dict(skips=[], ln=None, on_x=False, ws=ws),
                        'skip3': dict(skips=[s0, s1, s2], ln=None, on_x=False, ws=None)}
            for fname, f in flavours.items():
                nbytes = elem * a.batch * c * t * (2 + len(f['skips'])) + 4 * (c * (c // 100) * a.kernel + c)
                variants = (0, 1, 2, 3)
                times = {v: [] for v in variants}

                def run(v):
                    hip.grouped_conv1d_node(x, wp if v & hip.GC_WPERM else w, bias, f['skips'], y, t, 100, a.kernel, a.dilation,
                                            f['ln'], f['on_x'], False, f['ws'], v)
                for v in variants:
                    run(v)
                torch.cuda.synchronize()
                for _ in range(a.rounds):
                    for v in variants:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(a.reps):
                            run(v)
                        e1.record()
                        e1.synchronize()
                        times[v].append(e0.elapsed_time(e1) * 1e3 / a.reps)
                row = {'block': blk, 'C': c, 'T': t, 'dtype': dname, 'flavour': fname, 'MB': nbytes / 1e6}
                for v in variants:
                    us = statistics.median(times[v])
                    row[f'v{v}_us'] = round(us, 1)
                    row[f'v{v}_TBps'] = round(nbytes / us / 1e6, 2)
                out.append(row)
                print(json.dumps(row), flush=True)
    if a.json:
        pathlib.Path(a.json).write_text(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
