#!/usr/bin/env python3
"""Only the pipelined forward, for kernel traces of the small-batch regime:

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/ubench/small_batch_trace.py --batch 8
"""
import argparse
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import get_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--frames', type=int, default=1000)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--mode', choices=('async', 'forward'), default='async')
args = ap.parse_args()
torch.manual_seed(0)
model = get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0, gpu=0).eval()
x = torch.randn(args.batch, 80, args.frames, device='cuda:0')
with torch.no_grad():
    for phase, n in (('warm', 10), ('timed', args.steps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hs = [model.forward_async(x) if args.mode == 'async' else model(x) for _ in range(n)]
        for h in hs:
            if hasattr(h, 'result'):
                h.result()
        torch.cuda.synchronize()
        print(phase, round((time.perf_counter() - t0) / n * 1e3, 3), 'ms/step', flush=True)
