#!/usr/bin/env python3
"""Diagnostic (not a test): per-layer error of the HIP forward and of the fp32 CPU oracle against an fp64
evaluation of the same model, relative to each layer's scale.

    python tests/layer_noise.py [fixture tag]
    python tests/layer_noise.py --arch '[[0,1],[5,1,0],[2,0,1,1]]' --batch 2 --frames 258 --no-rnn --seed 77 --xseed 5
"""
import argparse
import json
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
from oracle import asr_oracle as oracle

ap = argparse.ArgumentParser()
ap.add_argument('tag', nargs='?', default=None)
ap.add_argument('--arch', default=None)
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--frames', type=int, default=200)
ap.add_argument('--no-rnn', action='store_true')
ap.add_argument('--mode', default='lively')
ap.add_argument('--seed', type=int, default=1235)
ap.add_argument('--xseed', type=int, default=0)
a = ap.parse_args()
if a.arch:
    todo = [('custom', json.loads(a.arch), not a.no_rnn, a.mode, a.batch, a.frames)]
else:
    todo = [c for c in cases.MODEL_CASES if c[0] == (a.tag or 'A_lively_b1_t500')]
rms = lambda v: float(v.double().pow(2).mean().sqrt())          # noqa: E731
for tag, arch, use_rnn, mode, b, t in todo:
    m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=a.seed, mode=mode)
    params = dict(m.state_dict())
    x = keyed_input(b, t, seed=a.xseed)
    t32, t64 = {}, {}
    oracle.asr_forward(params, arch, x, use_rnn=use_rnn, taps=t32)
    oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64, taps=t64)
    m = m.to('cuda:0').eval()
    with torch.no_grad():
        _, th = m.forward_with_taps(x.to('cuda:0'))
    print(f'{tag} {arch} b={b} t={t} rnn={use_rnn}: layer | scale | max err hip-f64 / scale | max err cpu32-f64 / scale | rms hip | rms cpu32 | rms ratio')
    for idx in sorted(t64):
        ref = t64[idx]
        scale = float(ref.abs().max()) + 1e-300
        eh = (th[idx].cpu().double() - ref)
        ec = (t32[idx].double() - ref)
        print(f'{idx:3d} {scale:10.3e} {float(eh.abs().max()) / scale:10.3e} {float(ec.abs().max()) / scale:10.3e} '
              f'{rms(eh) / scale:10.3e} {rms(ec) / scale:10.3e} {rms(eh) / max(rms(ec), 1e-300):6.2f}')
