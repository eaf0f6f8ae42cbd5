#!/usr/bin/env python3
"""Diagnostic: ONE SearchCell (block 2, C = 1000, 19 frames) forward + backward through the HIP autograd functions against ATen
autograd of the oracle's cell, at several input scales (uses the oracle)."""
import pathlib, sys
import torch
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
import cases
import nb_asr_amd as nb
from nb_asr_amd import autograd as nba
from nb_asr_amd.weights import keyed_fill_
from oracle import asr_oracle as oracle

arch = cases.ARCH_D
m = keyed_fill_(nb.get_model(arch, use_rnn=True, dropout_rate=0.0), seed=91, mode='lively').to('cuda:0').train()
names = oracle.arch_names(arch)
rms = lambda t: float(t.double().pow(2).mean().sqrt())
for idx in (13, 14):
    cell = m.model[idx]
    pre = f'model.{idx}.'
    state = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(pre)}
    for scale, seed in ((1.0, 0), (1.0, 1), (4.0, 2)):
        torch.manual_seed(seed)
        x = torch.randn(2, 1000, 19) * scale
        r = torch.randn(2, 1000, 19)
        params = {k: v.clone().double().requires_grad_(True) for k, v in state.items()}
        xr = x.clone().double().requires_grad_(True)
        (oracle.cell_forward(xr, names, params, pre) * r.double()).sum().backward()

        class Wrap(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.model = torch.nn.ModuleList([cell])
        xg = x.cuda().requires_grad_(True)
        m.zero_grad()
        y = nba.model_forward(Wrap(), xg)
        (y * r.cuda()).sum().backward()
        line = [f'cell {idx} scale {scale} seed {seed}: dx {rms(xg.grad.cpu().double() - xr.grad) / rms(xr.grad):.1e}']
        for key, p in cell.named_parameters():
            t = params[pre + key].grad
            line.append(f'{key.replace("nodes.", "n").replace(".op.conv.", ".").replace("norm_layer", "ln")} {rms(p.grad.cpu().double() - t) / (rms(t) + 1e-30):.1e}')
        print('  '.join(line), flush=True)
