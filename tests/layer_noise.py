#!/usr/bin/env python3
"""Diagnostic (not a test): per-layer error of the HIP forward and of the fp32 CPU oracle against an fp64
evaluation of the same model, relative to each layer's scale.  Usage: python tests/layer_noise.py [tag]"""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
from oracle import asr_oracle as oracle

want_tag = sys.argv[1] if len(sys.argv) > 1 else 'A_lively_b1_t500'
for tag, arch, use_rnn, mode, b, t in cases.MODEL_CASES:
    if tag != want_tag:
        continue
    m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=1235, mode=mode)
    params = dict(m.state_dict())
    x = keyed_input(b, t, seed=0)
    t32, t64 = {}, {}
    oracle.asr_forward(params, arch, x, use_rnn=use_rnn, taps=t32)
    oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64, taps=t64)
    m = m.to('cuda:0').eval()
    with torch.no_grad():
        _, th = m.forward_with_taps(x.to('cuda:0'))
    print(f'{tag}: layer | scale | max err hip-f64 / scale | max err cpu32-f64 / scale | rms hip | rms cpu32')
    for idx in sorted(t64):
        ref = t64[idx]
        scale = float(ref.abs().max()) + 1e-300
        eh = (th[idx].cpu().double() - ref).abs()
        ec = (t32[idx].double() - ref).abs()
        print(f'{idx:3d} {scale:10.3e} {float(eh.max()) / scale:10.3e} {float(ec.max()) / scale:10.3e} '
              f'{float((eh ** 2).mean().sqrt()) / scale:10.3e} {float((ec ** 2).mean().sqrt()) / scale:10.3e}')
