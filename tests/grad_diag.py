#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of loss.backward() through ASRModel against the fp64 oracle (uses the oracle: lives under
tests/).  usage: python tests/grad_diag.py"""
import pathlib, sys
import torch
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
from oracle import asr_oracle as oracle

arch, use_rnn = cases.ARCH_D, True
m = keyed_fill_(nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0), seed=91, mode='lively').to('cuda:0').train()
x = keyed_input(2, 37, seed=7)
state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
grads, r = {}, None
for dtype in (torch.float32, torch.float64):
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in state.items()}
    ref = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=dtype, differentiable=True)
    if r is None:
        r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5))
    (ref * r.to(dtype)).sum().backward()
    grads[dtype] = {k: p.grad for k, p in params.items() if p.grad is not None}
out = m(x.to('cuda:0'))
(out * r.to('cuda:0')).sum().backward()
rms = lambda t: float(t.double().pow(2).mean().sqrt())
for key, p in m.named_parameters():
    truth = grads[torch.float64].get(key)
    if truth is None or p.grad is None:
        print(f'{key:44s} missing: truth {truth is not None} hip {p.grad is not None}')
        continue
    size = rms(truth) + 1e-30
    print(f'{key:44s} hip {rms(p.grad.cpu().double() - truth) / size:9.2e}  fp32 oracle {rms(grads[torch.float32][key].double() - truth) / size:9.2e}')
