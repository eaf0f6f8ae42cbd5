#!/usr/bin/env python3
"""Nothing but W chains of pipelined forwards (model.forward_many), for a kernel trace: python tools/ubench/in_flight_run.py --batch 8 --ways 3 --steps 90"""
import argparse
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--ways', type=int, default=2)
ap.add_argument('--steps', type=int, default=90)
ap.add_argument('--tail-group', default='auto')
a = ap.parse_args()
tg = a.tail_group if a.tail_group == 'auto' else int(a.tail_group)
dev = torch.device('cuda', 0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
keyed_fill_(model, seed=1235, mode='lively')
model = model.to(dev).eval()
x = keyed_input(a.batch, 1000, seed=0).to(dev)
model.forward_many([x] * a.steps, in_flight=a.ways, tail_group=tg)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    model.forward_many([x] * a.steps, in_flight=a.ways, tail_group=tg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{a.ways} chains, tail group {tg}: {a.batch * a.steps / dt:8.0f} utterances/s ({1e3 * dt / a.steps:.3f} ms per step)', flush=True)
