#!/usr/bin/env python3
"""Same-process A/B of the round-2 node kernel (template over storage type, grouped_conv_impl.h) against the round-1 kernel
(tools/ubench/r1/grouped_conv_r1.hip, compiled into its own shared object at run time): interleaved rounds, one device.

    python tools/ubench/ab_gc_r1.py
"""
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

so = HERE / 'r1' / 'libgc_r1.so'
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-shared', f'-I{REPO}/include',
                f'-I{REPO}/nb_asr_amd/csrc', '-x', 'hip', str(HERE / 'r1' / 'grouped_conv_r1.hip'), str(REPO / 'nb_asr_amd/csrc/api.cpp'),
                '-o', str(so)], check=True)
r1 = ctypes.CDLL(str(so))
fn = r1.nbasr_grouped_conv1d_fused_stats
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                           ctypes.c_float, ctypes.c_void_p]
dev = torch.device('cuda', 0)
B, t = 64, 1000
rows = []
for blk, (c, stride) in enumerate(zip((600, 800, 1000, 1200), (1, 1, 2, 2))):
    t = (t + stride - 1) // stride
    ld = (t + 3) & ~3
    # two buffer sets larger than the last-level cache together, alternated, so that neither kernel is fed from it
    sets = []
    for _ in range(3):
        x = torch.randn(B, c, ld, device=dev) * 0.5
        x[:, :, t:] = 0
        sets.append((x, torch.empty_like(x)))
    w = torch.randn(c, c // 100, 5, device=dev) * 0.2
    bias = torch.randn(c, device=dev) * 0.1
    stats = torch.empty(B, 2, ld, device=dev)
    hip.channel_stats(sets[0][0], stats, t, 1e-3)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ln = hip.DeferredLN(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    ws = hip.grouped_stats_workspace(B, ld, 100, dev)
    stream = torch.cuda.current_stream().cuda_stream
    for flavour in ('plain', 'lnx', 'stats'):
        def new(i):
            x, y = sets[i % 3]
            hip.grouped_conv1d_node(x, w, bias, [], y, t, 100, 5, 1, (stats, gamma, beta) if flavour == 'lnx' else None, flavour == 'lnx', False,
                                    ws if flavour == 'stats' else None, 0)

        def old(i):
            x, y = sets[i % 3]
            rc = fn(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, None, None, y.data_ptr(), B, c, t, ld, 100, 5, 1,
                    ctypes.addressof(ln) if flavour == 'lnx' else None, int(flavour == 'lnx'), 0, None, ws.data_ptr() if flavour == 'stats' else None,
                    0.0, stream)
            assert rc == 0
        res = {'new': [], 'old': []}
        for f in (new, old):
            for i in range(3):
                f(i)
        torch.cuda.synchronize()
        for _ in range(9):
            for name, f in (('new', new), ('old', old)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(12):
                    f(i)
                e1.record()
                e1.synchronize()
                res[name].append(e0.elapsed_time(e1) * 1e3 / 12)
        nbytes = 4 * (2 * B * c * t + c * (c // 100) * 5 + c)
        row = {'block': blk, 'C': c, 'T': t, 'flavour': flavour, 'new_us': round(statistics.median(res['new']), 1), 'old_us': round(statistics.median(res['old']), 1),
               'new_TBps': round(nbytes / statistics.median(res['new']) / 1e6, 2), 'old_TBps': round(nbytes / statistics.median(res['old']) / 1e6, 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        # same results
        new(0); a = sets[0][1].clone(); old(0); torch.cuda.synchronize()
        assert torch.equal(a, sets[0][1]), 'round-1 and round-2 kernels disagree'
