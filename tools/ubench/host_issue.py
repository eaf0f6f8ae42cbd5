#!/usr/bin/env python3
"""Host-side issue time against device time of one forward, per batch size (is the small-batch regime launch-bound on the
host or on the device?).

    python tools/ubench/host_issue.py [--batches 8 16 32 64] [--frames 1000]
"""
import argparse
import json
import os
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import get_model  # noqa: E402

ARCH = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]       # bench.py's architecture (BASELINE configs[1]/[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, nargs='+', default=[8, 16, 32, 64])
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--steps', type=int, default=60)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = get_model(ARCH, use_rnn=True, dropout_rate=0.0, gpu=0).eval()
    for B in args.batches:
        x = torch.randn(B, 80, args.frames, device=dev)
        row = {'batch': B, 'frames': args.frames}
        for name, call in (('forward', lambda: model(x)), ('forward_async', lambda: model.forward_async(x)),
                           ('forward_graph', lambda: model.forward_graph(x))):
            with torch.no_grad():
                for _ in range(8):
                    r = call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hs = [call() for _ in range(args.steps)]
                t1 = time.perf_counter()
                for h in hs:
                    if hasattr(h, 'result'):
                        h.result()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            row[name] = {'host_issue_ms': round((t1 - t0) / args.steps * 1e3, 3), 'wall_ms': round((t2 - t0) / args.steps * 1e3, 3)}
        row['tape'] = os.environ.get('NBASR_TAPE', '1') != '0'
        row['tape_replays'] = sum(p.tape_replays for p in model._plans.values())
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
