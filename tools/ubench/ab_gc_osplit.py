#!/usr/bin/env python3
"""Default fp32 node kernel against the output-split variant (NBASR_GC_OSPLIT), per block of the benchmark configuration, as a
producer -> consumer chain through buffers larger than the last-level cache (= inside the model), interleaved rounds.

    python tools/ubench/ab_gc_osplit.py [--batches 64 8]
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
ap.add_argument('--kernel', type=int, default=5)
ap.add_argument('--dilation', type=int, default=1)
ap.add_argument('--lnx0', action='store_true', help='also LayerNorm on load without a skip input (the first node of a skip-free cell)')
args = ap.parse_args()
dev = torch.device('cuda', 0)
for B in args.batches:
    t = 1000
    for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
        t = (t + stride - 1) // stride
        ld = (t + 3) & ~3
        w = torch.randn(c, c // 100, args.kernel, device=dev) * 0.2
        bias = torch.randn(c, device=dev) * 0.1
        nbuf = max(4, int(600e6 / (B * c * ld * 4)) + 1)
        nbuf = min(nbuf, 24)
        bufs = [torch.randn(B, c, ld, device=dev) * 0.5 for _ in range(nbuf)]
        for b_ in bufs:
            b_[:, :, t:] = 0
        stats = torch.empty(B, 2, ld, device=dev)
        hip.channel_stats(bufs[0], stats, t, 1e-3)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        for flavour, n_skips in (('plain', 0), ('plain', 1), ('lnx', 1)) + ((('lnx', 0),) if args.lnx0 else ()):
            ln = (stats, gamma, beta) if flavour == 'lnx' else None

            def run(i, variant):
                src, dst = bufs[i % nbuf], bufs[(i + 1) % nbuf]
                skips = [bufs[(i + 2) % nbuf]] if n_skips and flavour != 'lnx' else ([src] if n_skips else [])
                hip.grouped_conv1d_node(src, w, bias, skips, dst, t, 100, args.kernel, args.dilation, ln, flavour == 'lnx', flavour == 'lnx' and bool(skips), None, variant)

            variants = (0, hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT)
            times = {v: [] for v in variants}
            for rnd in range(5):
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
            print(json.dumps({'k': args.kernel, 'd': args.dilation, 'batch': B, 'block': blk, 'C': c, 'T': t, 'flavour': flavour, 'skips': n_skips,
                              'default_us': round(statistics.median(times[0]), 1), 'osplit_us': round(statistics.median(times[hip.GC_OSPLIT]), 1),
                              'pipe_us': round(statistics.median(times[hip.GC_PIPE]), 1), 'pipe_osplit_us': round(statistics.median(times[hip.GC_PIPE | hip.GC_OSPLIT]), 1)}), flush=True)
        del bufs
