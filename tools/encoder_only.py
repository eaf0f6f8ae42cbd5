#!/usr/bin/env python3
"""How much of the pipelined step is the encoder alone?  Times the same architecture with and without the LSTM tail."""
import pathlib, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
ARCH = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = keyed_input(BATCH, 1000, seed=0).to('cuda:0')
for use_rnn in (False, True):
    m = nb.get_model(ARCH, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=1235, mode='lively')
    m = m.to('cuda:0').eval()
    with torch.no_grad():
        for mode in ('sequential', 'pipelined'):
            if mode == 'pipelined' and not use_rnn:
                continue
            for _ in range(5):
                y = m.forward_async(x).result() if mode == 'pipelined' else m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            pend = []
            for _ in range(n):
                if mode == 'pipelined':
                    pend.append(m.forward_async(x))
                    if len(pend) > 1:
                        pend.pop(0).result()
                else:
                    m(x)
            for p in pend:
                p.result()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            # host-only launch cost: how long the loop takes to ENQUEUE one forward
            t1 = time.perf_counter()
            y = m.forward_async(x) if (mode == 'pipelined') else m(x)
            enq = (time.perf_counter() - t1) * 1e3
            torch.cuda.synchronize()
            print(f'batch={BATCH} use_rnn={use_rnn} {mode}: {dt:.2f} ms/forward, enqueue of one forward {enq:.2f} ms', flush=True)
