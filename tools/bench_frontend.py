#!/usr/bin/env python3
"""Time of the HIP feature front-end (waveform -> normalised log-mel) at the benchmark shape: 64 utterances of 10 s."""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd.frontend import LogMelFrontend
DEV = 'cuda:0'
fe = LogMelFrontend(device=DEV)
wave = torch.randn(64, 159840, device=DEV) * 0.1            # -> 1000 frames
for _ in range(3):
    y = fe(wave)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = fe(wave); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
ms = statistics.median(ts)
flops = 2.0 * 64 * y.shape[2] * (402 * 400 + 80 * 204)
print(f'front-end B=64 x {wave.shape[1]} samples -> {tuple(y.shape)}: {ms * 1e3:.0f} us, {64 / ms * 1e3:.0f} utterances/s, {flops / ms / 1e9:.1f} TFLOP/s (DFT + mel GEMMs)')
