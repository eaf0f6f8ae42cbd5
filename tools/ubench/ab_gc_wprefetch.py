#!/usr/bin/env python3
"""Exploration for the next round: does issuing the scalar weight loads of channel ci + 1 before the FMAs of channel ci help the
pipelined, output-split fp32 node kernel?  tools/ubench/x1/gc_wprefetch.hip holds that kernel's plain flavour twice (WPF on / off), built
into its own shared object here; interleaved rounds in one process, three buffer sets in rotation (larger than the last-level cache),
outputs compared bit for bit, the library's own PIPE | OSPLIT variant timed beside them.

    python tools/ubench/ab_gc_wprefetch.py [--batches 64 8]
"""
import argparse
import ctypes
import json
import pathlib
import statistics
import subprocess
import sys

import torch

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
from nb_asr_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='+', default=[64, 8])
ap.add_argument('--rounds', type=int, default=30)
args = ap.parse_args()

so = HERE / 'x1' / 'libgc_wpf.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared', f'-I{REPO}/include',
                f'-I{REPO}/nb_asr_amd/csrc', '-x', 'hip', str(HERE / 'x1' / 'gc_wprefetch.hip'), '-o', str(so)], check=True)
x1 = ctypes.CDLL(str(so))
x1.x1_node.restype = ctypes.c_int
x1.x1_node.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream

for B in args.batches:
    t = 1000
    for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
        t = (t + stride - 1) // stride
        ld = (t + 3) & ~3
        sets = []
        for _ in range(3):
            x = torch.randn(B, c, ld, device=dev) * 0.5
            x[:, :, t:] = 0
            sets.append((x, torch.empty_like(x)))
        w = torch.randn(c, c // 100, 5, device=dev) * 0.2
        bias = torch.randn(c, device=dev) * 0.1

        def run(kind, i):
            x, y = sets[i % 3]
            if kind == 'lib':
                hip.grouped_conv1d_node(x, w, bias, [], y, t, 100, 5, 1, None, False, False, None, hip.GC_PIPE | hip.GC_OSPLIT)
            else:
                rc = x1.x1_node(int(kind == 'wpf'), x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, c, t, ld, stream)
                assert rc == 0, rc
            return y

        ref = run('lib', 0).clone()
        for kind in ('base', 'wpf'):
            assert torch.equal(run(kind, 0), ref), (kind, 'differs from the library kernel')
        times = {k: [] for k in ('lib', 'base', 'wpf')}
        for r in range(args.rounds + 3):
            for kind in times:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(6):
                    run(kind, i)
                e1.record()
                e1.synchronize()
                if r >= 3:
                    times[kind].append(e0.elapsed_time(e1) / 6 * 1000)
        row = {'batch': B, 'block': blk, 'channels': c, 'frames': t}
        row.update({k + '_us': round(statistics.median(v), 2) for k, v in times.items()})
        row['wpf_vs_base'] = round(row['wpf_us'] / row['base_us'], 4)
        print(json.dumps(row), flush=True)
