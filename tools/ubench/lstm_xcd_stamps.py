"""Per-frame phase times of lstm_xcd_kernel from in-kernel shader-clock stamps (diagnostic library built with -DNBASR_LX_STAMPS=1:
tools/ubench/build_lx_stamps.sh).  Wave 0 of every slice of utterance tile 0 stamps: 0 loop top, 1 flags all set, 2 quarter in LDS,
3 barrier passed, 4 sums in registers, 5 gates done, 6 pieces stored, drained, flag set, 7 h_out stored."""
import pathlib
import sys

import torch

REPO = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from nb_asr_amd import hip

hip.LIB_PATH = REPO / 'nb_asr_amd' / 'lib' / 'libnbasr_hip_stamps.so'
DEV = 'cuda:0'
frames, hidden = 250, 500
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
w_hh = (torch.rand(4 * hidden, hidden, device=DEV) * 2 - 1) * 0.049
p16 = hip.lstm_pack_whh16(w_hh)
gates = torch.randn(frames, b, 4 * hidden, device=DEV)
cell = torch.empty(b, hidden, device=DEV)
out = torch.empty(b, frames, hidden, device=DEV)
ws = hip.lstm_xcd_workspace(b, hidden, DEV)
for _ in range(3):
    hip.lstm_recurrence_xcd(gates, p16, cell, out, ws)
torch.cuda.synchronize()
hip.lstm_seq_status(ws)
points, nfr = 8, 256
n_tiles = (b + 15) // 16
tail = ws[ws.numel() - 32 * nfr * points * 8:].view(torch.int64).view(32, nfr, points).cpu().double()
st = tail[:, 20:240]                                       # steady state
names = ["flags polled (top -> all set)", "DMA of the quarter landed", "barrier", "gate loads issued + 48 MFMAs + sums", "gate math", "h split, gather, stores drained, flag", "h_out store + shift"]
d = st[:, :, 1:] - st[:, :, :-1]
period = (st[:, 1:, 0] - st[:, :-1, 0]).mean()
print(f'batch {b}: frame period {period:.0f} cycles (shader clock)')
for i, n in enumerate(names):
    print(f'  {n:28s} mean {d[:, :, i].mean():7.0f}   min over slices {d[:, :, i].mean(dim=1).min():7.0f}   max {d[:, :, i].mean(dim=1).max():7.0f}')
print(f'  loop tail -> next top        mean {(st[:, 1:, 0] - st[:, :-1, 7]).mean():7.0f}')
