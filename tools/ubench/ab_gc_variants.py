#!/usr/bin/env python3
"""Every variant of the fp32 node kernel per (taps, dilation, block, flavour, batch), as a producer -> consumer chain through buffers
larger than the last-level cache (= inside the model), interleaved rounds in one process.  The input of tools/make_gc_variant_table.py.

    python tools/ubench/ab_gc_variants.py [--batches 64 8] [--ops 5,1 5,2 7,1 7,2] > profiles/r03_gc_variants/<name>.jsonl

flavours: plain (no skip), skip (one skip input), lnx (LayerNorm on load of the main input, no skip), lnx+skip (and of skip0),
stats0 / stats (statistics epilogue -- the last node of a cell -- without / with one skip); variants with an output split have no
statistics flavour.
Row: {'k','d','batch','block','C','T','flavour', 'v<variant>': microseconds, ...}.
"""
import argparse
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='+', default=[64, 8])
ap.add_argument('--ops', nargs='+', default=['5,1', '5,2', '7,1', '7,2'])
ap.add_argument('--rounds', type=int, default=5)
args = ap.parse_args()
dev = torch.device('cuda', 0)
ALL = (0, hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING)
for B in args.batches:
    for op in args.ops:
        k, d = (int(v) for v in op.split(','))
        t = 1000
        for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
            t = (t + stride - 1) // stride
            ld = (t + 3) & ~3
            w = torch.randn(c, c // 100, k, device=dev) * 0.2
            bias = torch.randn(c, device=dev) * 0.1
            nbuf = min(max(4, int(600e6 / (B * c * ld * 4)) + 1), 24)
            bufs = [torch.randn(B, c, ld, device=dev) * 0.5 for _ in range(nbuf)]
            for b_ in bufs:
                b_[:, :, t:] = 0
            stats = torch.empty(B, 2, ld, device=dev)
            hip.channel_stats(bufs[0], stats, t, 1e-3)
            gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
            ws = hip.grouped_stats_workspace(B, ld, 100, dev)
            for flavour in ('plain', 'skip', 'lnx', 'lnx+skip', 'stats0', 'stats'):
                lnx = flavour.startswith('lnx')
                ln = (stats, gamma, beta) if lnx else None
                variants = tuple(v for v in ALL if not (flavour.startswith('stats') and v & hip.GC_OSPLIT))

                def run(i, variant):
                    src, dst = bufs[i % nbuf], bufs[(i + 1) % nbuf]
                    if flavour in ('skip', 'stats'):
                        skips = [bufs[(i + 2) % nbuf]]
                    else:
                        skips = [src] if flavour == 'lnx+skip' else []
                    hip.grouped_conv1d_node(src, w, bias, skips, dst, t, 100, k, d, ln, lnx, lnx and bool(skips),
                                            ws if flavour.startswith('stats') else None, variant)

                times = {v: [] for v in variants}
                for rnd in range(args.rounds):
                    for variant in variants:
                        for i in range(4):
                            run(i, variant)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        n = 40
                        e0.record()
                        for i in range(n):
                            run(i, variant)
                        e1.record()
                        torch.cuda.synchronize()
                        times[variant].append(e0.elapsed_time(e1) * 1e3 / n)
                row = {'k': k, 'd': d, 'batch': B, 'block': blk, 'C': c, 'T': t, 'flavour': flavour}
                row.update({f'v{v}': round(statistics.median(ts), 1) for v, ts in times.items()})
                print(json.dumps(row), flush=True)
            del bufs
