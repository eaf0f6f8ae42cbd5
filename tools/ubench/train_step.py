#!/usr/bin/env python3
"""Time of one training step (training-mode forward + CTC loss + loss.backward()) on the differentiable path.

    python tools/ubench/train_step.py [--batch 16 --frames 400]
"""
import argparse
import json
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import nb_asr_amd as nb  # noqa: E402
from nb_asr_amd import ctc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--frames', type=int, default=400)
ap.add_argument('--steps', type=int, default=3)
args = ap.parse_args()
torch.manual_seed(0)
model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0, gpu=0).train()
x = torch.randn(args.batch, 80, args.frames, device='cuda:0')
t_out = (args.frames + 3) // 4
targets = torch.randint(1, 49, (args.batch, 20), dtype=torch.int32, device='cuda:0')
tl = torch.full((args.batch,), 20, dtype=torch.int32, device='cuda:0')
ol = torch.full((args.batch,), t_out, dtype=torch.int32, device='cuda:0')
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
times = []
for step in range(args.steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = ctc.training_loss(model(x), ol, targets, tl)
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
with torch.no_grad():
    model.eval()
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    infer = (time.perf_counter() - t0) / 5
print(json.dumps({'batch': args.batch, 'frames': args.frames, 'train_step_ms': round(1e3 * min(times[1:]), 1), 'first_step_ms': round(1e3 * times[0], 1),
                  'inference_forward_ms': round(1e3 * infer, 2), 'loss': float(loss), 'peak_memory_GB': round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
