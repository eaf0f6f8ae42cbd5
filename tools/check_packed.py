#!/usr/bin/env python3
"""Accuracy + speed of the split-bf16 dense conv vs the fp32-MFMA kernel and an fp64 CPU reference (diagnostic)."""
import pathlib, statistics, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip
DEV = 'cuda:0'


def ref64(x, w, b, s):
    k = w.shape[-1]
    lp, rp = hip.pad_amounts(k, 1, s)
    y = F.conv1d(F.pad(x.double(), (lp, rp)), w.double(), b.double(), stride=s)
    return torch.clamp(torch.relu(y), max=20.0)


def run(cin, cout, tin, s, b, time_it):
    torch.manual_seed(cin * 7 + cout + tin)
    x = torch.randn(b, cin, tin)
    w = torch.randn(cout, cin, 8) * (2.0 / (cin * 8)) ** 0.5
    bias = torch.randn(cout) * 0.1
    tout = (tin + s - 1) // s
    xd = torch.zeros(b, cin, hip.round_up4(tin), device=DEV)
    xd[:, :, :tin] = x.to(DEV)
    wd, bd = w.to(DEV), bias.to(DEV)
    y32 = torch.full((b, cout, hip.round_up4(tout)), float('nan'), device=DEV)
    y16 = torch.full_like(y32, float('nan'))
    yh = torch.full_like(y32, float('nan'))
    hip.dense_conv1d_fused(xd, tin, wd, bd, (), y32, s)
    packed = hip.pack_dense_weights(wd, s)
    packed_h = hip.pack_dense_weights(wd, s, 'f16x2')
    amax = xd.abs().amax(dim=(1, 2))
    hip.dense_conv1d_fused_packed(xd, tin, packed, cout, 8, bd, (), y16, s)
    hip.dense_conv1d_fused_packed(xd, tin, packed_h, cout, 8, bd, (), yh, s, scheme='f16x2', x_absmax=amax)
    torch.cuda.synchronize()
    msg = f'{cin:5d}->{cout:5d} T={tin:5d} s={s} B={b:3d}: '
    if not time_it:
        want = ref64(x, w, bias, s)
        scale = float(want.abs().max())
        e32 = (y32[:, :, :tout].cpu().double() - want)
        e16 = (y16[:, :, :tout].cpu().double() - want)
        eh = (yh[:, :, :tout].cpu().double() - want)
        pad_ok = bool(torch.all(y16[:, :, tout:] == 0)) and bool(torch.all(yh[:, :, tout:] == 0))
        msg += (f'max|err|/scale fp32-mfma {float(e32.abs().max()) / scale:.2e} bf16x3 {float(e16.abs().max()) / scale:.2e} f16x2 {float(eh.abs().max()) / scale:.2e}  '
                f'rms fp32-mfma {float((e32 ** 2).mean().sqrt()) / scale:.2e} bf16x3 {float((e16 ** 2).mean().sqrt()) / scale:.2e} f16x2 {float((eh ** 2).mean().sqrt()) / scale:.2e} pad_ok={pad_ok}')
    else:
        for name, fn in (('fp32-mfma', lambda: hip.dense_conv1d_fused(xd, tin, wd, bd, (), y32, s)),
                         ('bf16x3', lambda: hip.dense_conv1d_fused_packed(xd, tin, packed, cout, 8, bd, (), y16, s)),
                         ('f16x2', lambda: hip.dense_conv1d_fused_packed(xd, tin, packed_h, cout, 8, bd, (), yh, s, scheme='f16x2', x_absmax=amax))):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            med = statistics.median(ts)
            msg += f'{name} {med * 1e3:8.1f} us {2.0 * b * tout * cout * cin * 8 / med / 1e9:6.1f} TF   '
    print(msg, flush=True)


if __name__ == '__main__':
    for cin, cout, tin, s, b in ((24, 40, 37, 1, 2), (24, 40, 37, 2, 2), (80, 600, 50, 1, 1), (136, 200, 131, 2, 1), (136, 200, 300, 1, 1),
                                 (600, 136, 140, 2, 2), (20, 33, 7, 2, 2), (8, 8, 1, 1, 1), (600, 800, 256, 1, 2), (1000, 1200, 260, 2, 1)):
        run(cin, cout, tin, s, b, False)
    for cin, cout, tin, s in ((80, 600, 1000, 1), (600, 800, 1000, 1), (800, 1000, 1000, 2), (1000, 1200, 500, 2), (600, 768, 1024, 1)):
        run(cin, cout, tin, s, 64, True)
