#!/usr/bin/env python3
"""Reproduces the slow state of two chains in flight after the plans have been re-created (what bench.py's strict leg does), and says
whether the HOST calls or the device got slow.   usage: python tools/ubench/collapse_repro.py [--clear 1] [--steps 50]"""
import argparse
import pathlib
import statistics
import sys
import threading
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

ap = argparse.ArgumentParser()
ap.add_argument('--clear', type=int, default=1, help='how many times the plan pool is cleared (and rebuilt) before the small-batch run')
ap.add_argument('--steps', type=int, default=50)
ap.add_argument('--ways', type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda', 0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
keyed_fill_(model, seed=1235, mode='lively')
model = model.to(dev).eval()
x64, x8 = keyed_input(64, 1000, seed=0).to(dev), keyed_input(8, 1000, seed=0).to(dev)
host = []
orig = model.forward


def timed_forward(x, *args, **kwargs):
    t0 = time.perf_counter()
    h = orig(x, *args, **kwargs)
    host.append((threading.get_ident(), time.perf_counter() - t0))
    return h


model.forward = timed_forward


def run(x, label):
    for _ in range(3):
        host.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.forward_many([x] * a.steps, in_flight=a.ways)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per = [d for _, d in host] or [0.0]
        print(f'{label}: {x.shape[0] * a.steps / dt:8.0f} utterances/s ({1e3 * dt / a.steps:.3f} ms per step); host time per forward: median {1e3 * statistics.median(per):.3f} ms, '
              f'max {1e3 * max(per):.3f} ms, sum per chain {1e3 * sum(per) / a.ways:.1f} ms of {1e3 * dt:.1f}', flush=True)


with torch.no_grad():
    model(x64)
run(x64, '64, fresh')
run(x8, ' 8, fresh')
for k in range(a.clear):
    model._plans.clear()
    with torch.no_grad():
        model(x64)
        model.forward_async(x64).result()
    run(x64, f'64, after clear {k + 1}')
    run(x8, f' 8, after clear {k + 1}')
print('plans:', len(model._plans))
