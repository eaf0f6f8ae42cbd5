#!/usr/bin/env python3
"""What does the pipelined LSTM tail cost the encoder's stream?  Times K back-to-back forward_async calls as bench.py does, once as
shipped and once with the recurrence launches left out of the tail (--skip: wrong logits, timing only).  The difference is what
the side stream's 250 dependent launches per forward take from the main stream (CU slots, dispatch).

usage: python tools/ubench/tail_cost.py [--skip] [--batch 64] [--steps 30]"""
import argparse
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb
from nb_asr_amd import executor
from nb_asr_amd.weights import keyed_fill_, keyed_input

ap = argparse.ArgumentParser()
ap.add_argument('--skip', action='store_true')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--lib', default=None, help='another build of the library (the -DNBASR_LX_TAIL_EXPERIMENT=1 twin: NBASR_LX_PRIO=3 no weight fetch, 5 empty launches)')
a = ap.parse_args()
if a.lib:
    from nb_asr_amd import hip as _hip
    _hip.LIB_PATH = pathlib.Path(a.lib).resolve()
if a.skip:
    executor.ForwardPlan._recurrence = lambda self, *args, **kw: None
dev = torch.device('cuda', 0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
keyed_fill_(model, seed=1235, mode='lively')
model = model.to(dev).eval()
x = keyed_input(a.batch, 1000, seed=0).to(dev)


def run():
    with torch.no_grad():
        hs = [model.forward_async(x) for _ in range(a.steps)]
        return [h.result() for h in hs][-1]


for _ in range(3):
    run()
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'skip' if a.skip else 'full'}: {a.batch * a.steps / dt:8.0f} utterances/s  ({1e3 * dt / a.steps:.3f} ms per step)", flush=True)
