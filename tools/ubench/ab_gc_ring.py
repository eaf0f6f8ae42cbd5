#!/usr/bin/env python3
"""Exploration (round 3): the fp32 node kernel with LDS-DMA-staged input windows (tools/ubench/x2/gc_ring.hip) against the library's
kernels, plain flavour (no skips, no LayerNorm on load, no statistics).  Interleaved rounds in one process, three buffer sets in
rotation (larger than the last-level cache), outputs compared bit for bit with the library kernel.

    python tools/ubench/ab_gc_ring.py [--batches 64 8] [--ops 5,1 7,2]
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
ap.add_argument('--rounds', type=int, default=20)
ap.add_argument('--ops', nargs='+', default=['5,1', '7,2'])
ap.add_argument('--wgs', type=int, nargs='+', default=[0], help='persistent grid: workgroups per CU (0: as many as the LDS admits)')
ap.add_argument('--pitch', type=int, default=4, help='row pitch rounded up to a multiple of this many frames (32 = 128-byte-aligned rows)')
args = ap.parse_args()

so = HERE / 'x2' / 'libgc_ring.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared', f'-I{REPO}/include',
                f'-I{REPO}/nb_asr_amd/csrc', '-x', 'hip', str(HERE / 'x2' / 'gc_ring.hip'), '-o', str(so)], check=True)
x2 = ctypes.CDLL(str(so))
x2.x2_node.restype = ctypes.c_int
x2.x2_node.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream

for B in args.batches:
    for op in args.ops:
        k, d = (int(v) for v in op.split(','))
        t = 1000
        for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
            t = (t + stride - 1) // stride
            ld = -(-t // args.pitch) * args.pitch
            cg = c // 100
            sets = []
            for _ in range(3):
                x = torch.randn(B, c, ld, device=dev) * 0.5
                x[:, :, t:] = 0
                sets.append((x, torch.empty_like(x)))
            w = torch.randn(c, cg, k, device=dev) * 0.2
            bias = torch.randn(c, device=dev) * 0.1
            lds_wgs = (160 * 1024) // (4 * cg * 1088)

            def run(kind, i):
                x, y = sets[i % 3]
                if kind == 'lib0':
                    hip.grouped_conv1d_node(x, w, bias, (), y, t, 100, k, d, None, False, False, None, 0)
                elif kind == 'lib3':
                    hip.grouped_conv1d_node(x, w, bias, (), y, t, 100, k, d, None, False, False, None, hip.GC_PIPE | hip.GC_OSPLIT)
                elif kind == 'libring':
                    hip.grouped_conv1d_node(x, w, bias, (), y, t, 100, k, d, None, False, False, None, hip.GC_RING)
                else:
                    mode, wgs = (0, 0) if kind == 'ring' else (1, int(kind[4:]) or lds_wgs)
                    rc = x2.x2_node(mode, x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, c, t, ld, k, d, wgs, stream)
                    assert rc == 0, (kind, rc)
                return y

            kinds = ['lib0', 'lib3', 'libring', 'ring'] + [f'pers{n}' for n in args.wgs]
            ref = run('lib0', 0).clone()
            for kind in kinds[1:]:
                y = run(kind, 0)
                torch.cuda.synchronize()
                assert torch.equal(y, ref), (kind, B, op, blk, 'differs from the library kernel', float((y - ref).abs().max()))
            times = {kk: [] for kk in kinds}
            for r in range(args.rounds + 3):
                for kind in kinds:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(6):
                        run(kind, i)
                    e1.record()
                    e1.synchronize()
                    if r >= 3:
                        times[kind].append(e0.elapsed_time(e1) / 6 * 1000)
            nbytes = 4.0 * (2 * B * c * t + c * cg * k + c)
            row = {'batch': B, 'op': op, 'block': blk, 'channels': c, 'frames': t, 'ld': ld}
            for kk, v in times.items():
                us = statistics.median(v)
                row[kk + '_us'] = round(us, 2)
                row[kk + '_TBs'] = round(nbytes / us / 1e6, 2)
            print(json.dumps(row), flush=True)
