#!/usr/bin/env python3
"""Several forwards in flight on several streams of one device (one thread per stream): are the logits those of a lone forward?
usage: python tools/ubench/in_flight_check.py [--batch 8] [--ways 2] [--steps 30] [--plain]"""
import argparse
import pathlib
import sys
import threading

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--ways', type=int, default=2)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--plain', action='store_true', help='model(x) instead of forward_async')
ap.add_argument('--rounds', type=int, default=4)
ap.add_argument('--overlap-main', action='store_true', help='the main thread enqueues its own chain on the default stream first and does not wait')
ap.add_argument('--main-stream', action='store_true', help='with --overlap-main: the main thread uses a stream of its own instead of the default stream')
a = ap.parse_args()
dev = torch.device('cuda', 0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
keyed_fill_(model, seed=1235, mode='lively')
model = model.to(dev).eval()
xs = [keyed_input(a.batch, 1000, seed=i).to(dev) for i in range(a.ways)]
with torch.no_grad():
    want = [model(x).clone() for x in xs]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(a.ways)]
bad = [[] for _ in range(a.ways)]
main_stream = torch.cuda.Stream(device=dev)


def worker(i):
    with torch.no_grad(), torch.cuda.stream(streams[i]):
        if a.plain:
            outs = [model(xs[i]) for _ in range(a.steps)]
        else:
            hs = [model.forward_async(xs[i]) for _ in range(a.steps)]
            outs = [h.result() for h in hs]
        streams[i].synchronize()
        for k, o in enumerate(outs):
            if not torch.equal(o, want[i]):
                bad[i].append((k, float((o - want[i]).abs().max())))


for r in range(a.rounds):
    for b in bad:
        b.clear()
    main_outs = None
    if a.overlap_main:
        with torch.no_grad(), torch.cuda.stream(main_stream if a.main_stream else torch.cuda.default_stream(dev)):
            hs = [model.forward_async(xs[0]) for _ in range(a.steps)]
            main_outs = [h.result() for h in hs]
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(a.ways)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if main_outs is not None:
        torch.cuda.synchronize()
        nb_bad = [(k, float((o - want[0]).abs().max())) for k, o in enumerate(main_outs) if not torch.equal(o, want[0])]
        print(f'   main thread: {len(nb_bad)} of {a.steps} differ' + (f' (first step {nb_bad[0][0]}, max |diff| {max(v for _, v in nb_bad):.3g})' if nb_bad else ''))
    print(f'round {r}: ' + '  '.join(f'way {i}: {len(b)} of {a.steps} differ' + (f' (first step {b[0][0]}, max |diff| {max(v for _, v in b):.3g})' if b else '') for i, b in enumerate(bad)), flush=True)
print('plans in the pool:', len(model._plans))
