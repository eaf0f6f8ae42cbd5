#!/usr/bin/env python3
"""Times the validation decode kernels at the bench shape: log-probabilities (B, T/4, 49).
usage: python tools/bench_decode.py [--batch 64] [--frames 1000] [--iters 20]"""
import argparse
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from nb_asr_amd import ctc  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    out = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return statistics.median(out), min(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    t_out = (a.frames + 3) // 4
    gen = torch.Generator().manual_seed(0)
    for sharp, name in ((1.0, 'flat (untrained model)'), (6.0, 'peaked (trained model)')):
        logits = torch.randn(a.batch, t_out, 49, generator=gen) * sharp
        logits[:, ::2, 0] += 2.0 * sharp
        logits = logits.to(dev)
        lp = ctc.log_softmax(logits)
        out_len = torch.full((a.batch,), t_out, dtype=torch.int32)
        targets = torch.randint(1, 49, (a.batch, 60), generator=gen, dtype=torch.int32).to(dev)
        targets_len = torch.full((a.batch,), 60, dtype=torch.int32)
        for width in (1, 12, 32):
            med, mn = timeit(lambda: ctc.beam_decode(lp, out_len, beam_width=width), a.iters)
            print(f'{name}: beam search B={a.batch} T\'={t_out} width={width:2d}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})  '
                  f'{med * 1e3 / t_out:6.2f} us/frame')
        beams, _, lens = ctc.beam_decode(lp, out_len)
        hyp, hyp_len = beams[:, 0].contiguous(), lens[:, 0].contiguous()
        table = ctc.fold_table().to(dev)
        med, mn = timeit(lambda: ctc.error_rates(hyp, hyp_len, targets, targets_len, table=table), a.iters)
        print(f'{name}: fold + error rate, hyp {int(hyp_len.float().mean())} vs ref 60 tokens: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})')
        med, mn = timeit(lambda: ctc.greedy_decode(logits, None), a.iters)
        print(f'{name}: log_softmax + greedy decode (with the copy of the tokens to the host): {med * 1e3:8.1f} us')
        med, mn = timeit(lambda: ctc.ctc_loss(lp, out_len, targets, targets_len), a.iters)
        print(f'{name}: CTC loss value (60 labels per utterance): {med * 1e3:8.1f} us')
        med, mn = timeit(lambda: ctc.decode_per(lp, out_len, targets, targets_len), a.iters)
        print(f'{name}: Trainer.decode as a whole (beam 12 -> fold 39 -> error rate -> mean): {med * 1e3:8.1f} us')


if __name__ == '__main__':
    main()
