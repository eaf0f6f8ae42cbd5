#!/usr/bin/env python3
"""Error model of the fp16-split dense convolution (csrc/gemm_conv_split.hip), evaluated on the CPU in float64.

Why the round-2 default path sat at 1.2-1.4 x the CPU conv's error against fp64 (and 1.55-1.8 x on five no-LSTM models of the
150-case sweep), and what the round-3 accumulation does about it.  Separates three contributions for one
conv layer (LayerNorm-like input, He-uniform weights, both range-normalised by powers of two as the kernel does):

  * the SPLIT itself: v = hi + lo with two fp16 terms (23 bits + sign) and the dropped lo*lo product -- exact accumulation;
  * a single running sum per output, one fp32 rounding per 32-k MFMA (rounds 1-2; hi*hi and the cross terms in two register sets);
  * the two-level blocked sum of round 3: every channel group (16 channels x 8 taps = 4 k-blocks, the three products of a
    k-block smallest first into ONE accumulator) summed from zero, then added to the total.

The hardware rounds more often than once per MFMA (measured: conv 1 injects 1.47 x the CPU conv's noise where this model's
single chain gives 1.0 x; two roundings per MFMA fit that), which lengthens both chains alike.  usage: python tools/split_accumulation_model.py [c_in c_out frames]
"""
import sys

import torch

F = torch.nn.functional


def split16(v):
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)              # unscaled residual; fp16 subnormals are exact to 2^-24
    return hi.double(), lo.double()


def rms(a):
    return float(a.pow(2).mean().sqrt())


def main():
    cin, cout, frames = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (608, 800, 72)
    assert cin % 16 == 0
    k = 8
    torch.manual_seed(0)
    x = torch.randn(1, cin, frames)
    w = (torch.rand(cout, cin, k) * 2 - 1) * (6 / (cin * k)) ** 0.5
    xs = x * 2.0 ** (14 - int(torch.floor(torch.log2(x.abs().max()))))
    ws = w * 2.0 ** 13 / 2.0 ** torch.floor(torch.log2(w.abs().amax(dim=(1, 2), keepdim=True)))
    truth = F.conv1d(xs.double(), ws.double())
    sc = rms(truth)
    print(f'conv {cin} -> {cout}, k = 8, {truth.shape[-1]} output frames; errors are RMS against fp64, relative to the output RMS')
    print(f'  fp32 CPU conv (the reference\'s arithmetic)          {rms(F.conv1d(xs, ws).double() - truth) / sc:.3e}')
    xh, xl = split16(xs)
    wh, wl = split16(ws)
    s3 = F.conv1d(xh, wh) + F.conv1d(xh, wl) + F.conv1d(xl, wh)
    print(f'  split alone (3 products, exact accumulation)        {rms(s3 - truth) / sc:.3e}')
    print(f'  ... with the lo*lo product as well                   {rms(s3 + F.conv1d(xl, wl) - truth) / sc:.3e}')

    def unfold(t):
        return t[0].unfold(1, k, 1).permute(1, 0, 2).contiguous()            # (frames_out, c_in, taps)

    def kblocks(xu, wt):                                                      # exact sums of 32 k = 16 channels x 2 taps
        n = xu.shape[0]
        xb = xu.reshape(n, cin // 16, 16, 4, 2)
        wb = wt.reshape(cout, cin // 16, 16, 4, 2)
        return torch.einsum('tgcpq,ogcpq->togp', xb, wb).reshape(n, cout, -1)

    xuh, xul = unfold(xh), unfold(xl)
    b_hh, b_hl, b_lh = kblocks(xuh, wh), kblocks(xuh, wl), kblocks(xul, wh)
    tr = truth[0].t()

    def chain(blocks):
        acc = torch.zeros(blocks[0].shape[:-1], dtype=torch.float32)
        for i in range(blocks[0].shape[-1]):
            for b in blocks:
                acc = (acc.double() + b[..., i]).float()
        return acc.double()

    old = chain([b_hh]) + chain([(b_hl + b_lh) * 2048]) / 2048
    print(f'  rounds 1-2: one running sum, hi*hi | cross terms    {rms(old - tr) / sc:.3e}')
    for per_block in (4, 8, 16):
        tot = torch.zeros(b_hh.shape[:-1], dtype=torch.float32)
        for s in range(0, b_hh.shape[-1], per_block):
            part = chain([b[..., s:s + per_block] for b in (b_lh, b_hl, b_hh)])
            tot = (tot.double() + part).float()
        tag = ' (the kernel: one channel group)' if per_block == 4 else ''
        print(f'  round 3: blocks of {per_block:2d} k-blocks, one accumulator      {rms(tot.double() - tr) / sc:.3e}{tag}')


if __name__ == '__main__':
    main()
