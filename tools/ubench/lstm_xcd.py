"""Time the three recurrence forms at the model's size (T' = 250, H = 500): one launch per frame (graph-replayed), the round-4 resident grid
(lstm_seq_kernel: fp32 MFMA, chip-wide exchange), the round-6 XCD-local resident form (fp16-pair MFMA).  us per frame, HIP events."""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip

DEV = 'cuda:0'
frames, hidden = 250, 500
torch.manual_seed(0)
w_hh = (torch.rand(4 * hidden, hidden, device=DEV) * 2 - 1) * 0.049
p32, p16 = hip.lstm_pack_whh(w_hh), hip.lstm_pack_whh16(w_hh)
for b in [int(v) for v in (sys.argv[1:] or (8, 16, 32, 64, 128))]:
    gates = torch.randn(frames, b, 4 * hidden, device=DEV)
    cell = torch.empty(b, hidden, device=DEV)
    out = {k: torch.empty(b, frames, hidden, device=DEV) for k in ('packed', 'seq', 'xcd')}
    ws_x = hip.lstm_xcd_workspace(b, hidden, DEV)
    nb = hip.lstm_seq_workspace_bytes(b, hidden, DEV)
    ws_s = torch.empty(nb, dtype=torch.uint8, device=DEV) if nb else None
    forms = {'packed': lambda: hip.lstm_recurrence_packed(gates, p32, cell, out['packed']),
             'xcd': lambda: hip.lstm_recurrence_xcd(gates, p16, cell, out['xcd'], ws_x)}
    if ws_s is not None:
        forms['seq'] = lambda: hip.lstm_recurrence_seq(gates, p32, cell, out['seq'], ws_s)
    res = {}
    for name, fn in forms.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    hip.lstm_seq_status(ws_x)
    d = (out['xcd'] - out['packed']).abs().max().item()
    print(f'batch {b:4d}: ' + '  '.join(f'{k} {v:.3f} ms ({1e3 * v / frames:.2f} us/frame)' for k, v in res.items()) + f'  max|xcd - packed| {d:.2e}', flush=True)
