#!/usr/bin/env python3
"""Do two per-frame LSTM recurrence chains on two high-priority streams interleave on the hardware?  (round 4, small batch: the side
stream's chain of 250 dependent launches is the critical path of a pipelined step at 8 utterances per GPU.)
python tools/ubench/two_chains.py [batch=8] [frames=250]"""
import pathlib, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
H, DEV = 500, 'cuda:0'
torch.manual_seed(0)
whh = hip.lstm_pack_whh(torch.randn(4 * H, H, device=DEV) * 0.04)
slots = [(torch.randn(T, B, 4 * H, device=DEV), torch.zeros(B, H, device=DEV), torch.empty(B, T, H, device=DEV)) for _ in range(2)]
streams = [torch.cuda.Stream(device=DEV, priority=-1) for _ in range(2)]
# something for the main stream to chew on meanwhile (stands in for the encoder): a big elementwise chain
junk = torch.randn(64, 1200, 1000, device=DEV)


def chains(n, two, busy):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        s = streams[i % 2 if two else 0]
        g, c, h = slots[i % 2]
        with torch.cuda.stream(s):
            hip.lstm_recurrence_packed(g, whh, c, h)
        if busy:
            junk.mul_(1.0001)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for busy in (False, True):
    for two in (False, True, False, True):
        chains(4, two, busy)
        print(f'B={B} T={T} main stream busy={busy} two side streams={two}: {chains(20, two, busy):.3f} ms per chain', flush=True)
