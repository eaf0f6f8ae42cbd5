#!/usr/bin/env python3
"""Forward time of a few architectures at B=64, T=1000 (sequential forwards): python tools/arch_timing.py"""
import pathlib, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
ARCHS = {'conv5 x3, no skips (bench)': [[1, 0], [1, 0, 0], [1, 0, 0, 0]],
         'linear x3, no skips': [[0, 0], [0, 0, 0], [0, 0, 0, 0]],
         'conv7 / conv7d2 / conv5d2, all skips (config 4)': [[3, 1], [4, 1, 1], [2, 1, 1, 1]],
         'linear, conv5, zero + skips': [[0, 1], [1, 0, 1], [5, 1, 0, 1]]}
x = keyed_input(64, 1000, seed=0).to('cuda:0')
for name, arch in ARCHS.items():
    m = keyed_fill_(nb.get_model(arch, use_rnn=True, dropout_rate=0.0), seed=1235, mode='lively').to('cuda:0').eval()
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m(x)
        torch.cuda.synchronize()
    print(f'{name:50s} {(time.perf_counter() - t0) / 10 * 1e3:7.2f} ms/forward', flush=True)
