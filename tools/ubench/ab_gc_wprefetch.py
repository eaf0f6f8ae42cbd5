#!/usr/bin/env python3
"""Exploration for the next round: does issuing the scalar weight loads of channel ci + 1 before the FMAs of channel ci help the
pipelined, output-split fp32 node kernel?  tools/ubench/x1/gc_wprefetch.hip holds that kernel's plain flavour twice (WPF on / off), built
into its own shared object here; interleaved rounds in one process, three buffer sets in rotation (larger than the last-level cache),
outputs compared bit for bit, the library's own PIPE | OSPLIT variant timed beside them; `gen` = the same loop inside the library kernel's
generic prologue / epilogue (optional skips, LayerNorm on the first, ragged vote) -- what generality costs a short-lived wave; `nsk` = the
epilogue instantiated per number of skips, branch-free; `coop` = the two waves of a group share their window loads through LDS.
Each with 0, 1 and 2 skip inputs.

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
ap.add_argument('--skips', type=int, nargs='+', default=[0, 1, 2])
args = ap.parse_args()

so = HERE / 'x1' / 'libgc_wpf.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared', f'-I{REPO}/include',
                f'-I{REPO}/nb_asr_amd/csrc', '-x', 'hip', str(HERE / 'x1' / 'gc_wprefetch.hip'), '-o', str(so)], check=True)
x1 = ctypes.CDLL(str(so))
x1.x1_node.restype = ctypes.c_int
x1.x1_node.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
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

        cg = c // 100
        # the cooperative kernel's weight layout: [group][half][stage][co][channel of the stage][tap]
        w_coop = w.view(100, 2, cg // 2, cg // 2, 2, 5).permute(0, 1, 3, 2, 4, 5).contiguous()
        skip_sets = [torch.randn(B, c, ld, device=dev) for _ in range(2)]
        for sk in skip_sets:
            sk[:, :, t:] = 0

        def run(kind, i, n_skips=0):
            x, y = sets[i % 3]
            skips = skip_sets[:n_skips]
            if kind == 'lib':
                hip.grouped_conv1d_node(x, w, bias, skips, y, t, 100, 5, 1, None, False, False, None, hip.GC_PIPE | hip.GC_OSPLIT)
            else:
                ptr = [sk.data_ptr() for sk in skips] + [None] * (2 - n_skips)
                rc = x1.x1_node({'base': 0, 'wpf': 1, 'gen': 2, 'nsk': 3, 'coop': 4}[kind], x.data_ptr(), (w_coop if kind == 'coop' else w).data_ptr(), bias.data_ptr(), ptr[0], ptr[1],
                                y.data_ptr(), B, c, t, ld, stream)
                assert rc == 0, rc
            return y

        for n_skips in args.skips:
            kinds = ('lib', 'base', 'wpf', 'gen', 'nsk', 'coop') if n_skips == 0 else ('lib', 'gen', 'nsk')
            ref = run('lib', 0, n_skips).clone()
            for kind in kinds[1:]:
                assert torch.equal(run(kind, 0, n_skips), ref), (kind, n_skips, 'differs from the library kernel')
            times = {k: [] for k in kinds}
            for r in range(args.rounds + 3):
                for kind in times:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(6):
                        run(kind, i, n_skips)
                    e1.record()
                    e1.synchronize()
                    if r >= 3:
                        times[kind].append(e0.elapsed_time(e1) / 6 * 1000)
            row = {'batch': B, 'block': blk, 'channels': c, 'frames': t, 'skips': n_skips}
            row.update({k + '_us': round(statistics.median(v), 2) for k, v in times.items()})
            row['nsk_vs_gen'] = round(row['nsk_us'] / row['gen_us'], 4)
            if 'coop_us' in row:
                row['coop_vs_base'] = round(row['coop_us'] / row['base_us'], 4)
            print(json.dumps(row), flush=True)
