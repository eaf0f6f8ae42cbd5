#!/usr/bin/env python3
"""Logits error of the three dense-conv modes against the fp64 evaluation and the reference logits of tests/golden
(diagnostic; run on a GPU box): worst |err| / (atol + rtol |want|) with rtol 1e-4, atol 1e-5."""
import os, pathlib, sys
import numpy as np
import torch
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
fx = np.load(root / 'tests' / 'golden' / 'model_fixtures.npz')
for tag, arch, use_rnn, mode, b, t in cases.MODEL_CASES:
    line = f'{tag:28s} ref-noise {float(fx[tag + "/ref_noise_ratio"]):5.2f} |'
    for dense in ('f32', 'bf16x3', 'auto'):
        os.environ['NBASR_DENSE_MODE'] = dense
        m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
        keyed_fill_(m, seed=1235, mode=mode)
        m = m.to('cuda:0').eval()
        with torch.no_grad():
            y = m(keyed_input(b, t, seed=0).to('cuda:0')).cpu()
        r64 = cases.worst_ratio(y, torch.from_numpy(fx[tag + '/logits_f64']), 1e-4, 1e-5)
        rref = cases.worst_ratio(y, torch.from_numpy(fx[tag + '/logits']), 1e-4, 1e-5)
        line += f' {dense}: vs-f64 {r64:5.2f} vs-ref {rref:5.2f} |'
    print(line, flush=True)
