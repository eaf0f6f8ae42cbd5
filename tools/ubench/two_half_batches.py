#!/usr/bin/env python3
"""Experiment: does running the batch as TWO half-batches on two streams (two threads, two plans) beat one forward of the whole batch?
A kernel's drain on one stream would overlap the other stream's steady state; results are bit-identical either way (batch invariance).

usage: python tools/ubench/two_half_batches.py [--batch 64] [--steps 20]"""
import argparse
import pathlib
import sys
import threading
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--ways', type=int, default=2)
ap.add_argument('--full', action='store_true', help='every way runs the WHOLE batch (two batches in flight on two streams) instead of a slice of it')
ap.add_argument('--plain-first', type=int, default=0, help='that many plain model(x) calls (the resident recurrence) before everything else')
ap.add_argument('--events', action='store_true', help='an event behind every result, waited for by the main thread (as bench.py does)')
a = ap.parse_args()
dev = torch.device('cuda', 0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
keyed_fill_(model, seed=1235, mode='lively')
model = model.to(dev).eval()
x = keyed_input(a.batch, 1000, seed=0).to(dev)


if a.plain_first:
    with torch.no_grad():
        for _ in range(a.plain_first):
            model(x)
    torch.cuda.synchronize()


def whole():
    with torch.no_grad():
        hs = [model.forward_async(x) for _ in range(a.steps)]
        return [h.result() for h in hs][-1]


parts = [x.clone() if a.full else x[i * a.batch // a.ways:(i + 1) * a.batch // a.ways].contiguous() for i in range(a.ways)]
from nb_asr_amd import streams as stream_picker
streams = stream_picker.chain_streams(dev, a.ways)          # streams on different hardware queues, their tail streams chosen too (streams.py)
outs = [None] * a.ways


def worker(i):
    with torch.no_grad(), torch.cuda.stream(streams[i]):
        hs = [model.forward_async(parts[i]) for _ in range(a.steps)]
        if a.events:
            evs = []
            for h in hs:
                o = h.result()
                ev = torch.cuda.Event()
                ev.record(streams[i])
                evs.append((o, ev))
            outs[i] = evs
            return
        outs[i] = [h.result() for h in hs][-1]
        streams[i].synchronize()


def split():
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(a.ways)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if a.events:
        here, last = torch.cuda.current_stream(dev), None
        for evs in outs:
            for o, ev in evs:
                here.wait_event(ev)
                o.record_stream(here)
                last = o
        return last
    return outs[0] if a.full else torch.cat(outs, 0)


for fn in (whole, split, whole, split):
    fn()
torch.cuda.synchronize()
for name, fn in (('whole', whole), ('split', split), ('whole', whole), ('split', split)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = a.batch * a.steps * (a.ways if (a.full and name == 'split') else 1)
    print(f'{name}: {n / dt:8.0f} utterances/s  ({1e3 * dt / a.steps:.3f} ms per {a.batch})', flush=True)
print('bit-equal', torch.equal(whole(), split()))
