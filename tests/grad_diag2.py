#!/usr/bin/env python3
"""Diagnostic: the grouped node op's backward for every (channels per group, taps, dilation) against ATen autograd (uses the oracle)."""
import pathlib, sys
import torch
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
from nb_asr_amd import autograd as nba
from oracle import asr_oracle as oracle

rms = lambda t: float(t.double().pow(2).mean().sqrt())
for groups, b, t in ((100, 2, 19), (4, 2, 37)):
    for cg in (6, 8, 10, 12):
        for k, d in ((5, 1), (5, 2), (7, 1), (7, 2)):
            torch.manual_seed(cg * 100 + k * 10 + d)
            c = cg * groups
            x = torch.randn(b, c, t) * 2.0
            w = torch.randn(c, cg, k) * 0.3
            bias = torch.randn(c) * 0.2
            r = torch.randn(b, c, t)
            xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
            (oracle.pad_conv_relu(xr, wr, br, d, 1, groups) * r).sum().backward()
            xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
            (nba.grouped_pad_conv_relu(xg, wg, bg, groups, k, d) * r.cuda()).sum().backward()
            e = [rms(g.grad.cpu() - ref.grad) / (rms(ref.grad) + 1e-30) for g, ref in ((xg, xr), (wg, wr), (bg, br))]
            flag = '  <-----' if max(e) > 1e-4 else ''
            print(f'groups {groups:3d} t {t:2d} cg {cg:2d} k{k} d{d}: dx {e[0]:.1e} dw {e[1]:.1e} db {e[2]:.1e}{flag}', flush=True)
