#!/usr/bin/env python3
"""What the fp32 node kernel's time is made of: the round-1 kernel source built three ways -- as is, without its FMAs (loads +
stores only), without its loads (arithmetic + stores only) -- timed per block as a producer -> consumer chain through buffers
larger than the last-level cache.  If memory time and arithmetic time overlapped perfectly the full kernel would take the larger
of the two.

    python tools/ubench/gc_phases.py
"""
import ctypes
import json
import pathlib
import subprocess
import sys

import torch

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
from nb_asr_amd import hip  # noqa: E402  (only for the library's error helper and stream plumbing)

libs = {}
for name, flags in (('full', []), ('no_fma', ['-DGC_EXP_NOFMA=1']), ('no_load', ['-DGC_EXP_NOLOAD=1'])):
    so = HERE / 'r1' / f'libgc_{name}.so'
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared', f'-I{REPO}/include',
                    f'-I{REPO}/nb_asr_amd/csrc', *flags, '-x', 'hip', str(HERE / 'r1' / 'grouped_conv_r1.hip'), str(REPO / 'nb_asr_amd/csrc/api.cpp'),
                    '-o', str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    fn = lib.nbasr_grouped_conv1d_fused_stats
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                               ctypes.c_float, ctypes.c_void_p]
    libs[name] = fn

dev = torch.device('cuda', 0)
B, t = 64, 1000
for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
    t = (t + stride - 1) // stride
    ld = (t + 3) & ~3
    w = torch.randn(c, c // 100, 5, device=dev) * 0.2
    bias = torch.randn(c, device=dev) * 0.1
    nbuf = min(max(4, int(700e6 / (B * c * ld * 4)) + 1), 16)
    bufs = [torch.randn(B, c, ld, device=dev) * 0.5 for _ in range(nbuf)]
    stream = torch.cuda.current_stream().cuda_stream
    row = {'block': blk, 'C': c, 'T': t, 'MB': round(2 * B * c * ld * 4 / 1e6, 1)}
    for name, fn in libs.items():
        def run(i):
            rc = fn(bufs[i % nbuf].data_ptr(), w.data_ptr(), bias.data_ptr(), None, None, None, bufs[(i + 1) % nbuf].data_ptr(), B, c, t, ld, 100, 5, 1,
                    None, 0, 0, None, None, 0.0, stream)
            assert rc == 0
        best = 1e9
        for rnd in range(4):
            for i in range(4):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 40
            e0.record()
            for i in range(n):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        row[f'{name}_us'] = round(best, 1)
    print(json.dumps(row), flush=True)
    del bufs
