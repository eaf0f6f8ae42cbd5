#!/usr/bin/env python3
"""Per-workgroup phase stamps of the fused cell (grouped_cell.hip) -> per-phase durations and per-CU timelines.

Needs a library built with the stamps compiled in:
    NBASR_EXTRA_CXXFLAGS=-DNBASR_CELL_STAMPS=1 python -c "import nb_asr_amd.build as b; b.build_library(force=True)"
    python tools/gpu/cell_stamps.py [batch=64] [block=0..3] [--lib nb_asr_amd/lib/libnbasr_hip_cstamps.so]
Wave 0 of every workgroup records the 100 MHz clock at its phase boundaries into the buffer whose address the launcher reads from
NBASR_CELL_STAMPS.  The round-4 record is profiles/r04_cell_phase_stamps.txt."""
import os, sys, pathlib
import numpy as np
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from nb_asr_amd import hip
if '--lib' in sys.argv:            # tools/ubench/build_cell_stamps.sh: a stamped twin beside the shipped library
    i = sys.argv.index('--lib')
    hip.LIB_PATH = pathlib.Path(sys.argv[i + 1]).resolve()
    del sys.argv[i:i + 2]
DEV = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c, tt = ((600, 1000), (800, 1000), (1000, 500), (1200, 250))[blk]
k, d = 5, 1
ld = hip.round_up4(tt)
torch.manual_seed(0)
bufs = [torch.randn(b, c, ld, device=DEV) * 1.5 for _ in range(6)]
nodes = [(hip.pack_grouped_weights(torch.randn(c, c // 100, k, device=DEV) * 0.3, 100), torch.randn(c, device=DEV) * 0.2, k, d) for _ in range(3)]
stats = torch.empty(b, 2, ld, device=DEV)
hip.channel_stats(bufs[0], stats, tt, 1e-3)
ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
ws = hip.grouped_stats_workspace(b, ld, 100, DEV)
gpp = hip.grouped_cell_fits(c, ld, 100)
nwg = ((100 + gpp - 1) // gpp) * b
stamps = torch.zeros(nwg, 16, dtype=torch.int64, device=DEV)
for i in range(4):
    hip.grouped_cell_fused(bufs[i % 6], nodes, 0, bufs[(i + 1) % 6], tt, 100, ln, ws)
torch.cuda.synchronize()
os.environ['NBASR_CELL_STAMPS'] = hex(stamps.data_ptr())
hip.grouped_cell_fused(bufs[4], nodes, 0, bufs[5], tt, 100, ln, ws)
torch.cuda.synchronize()
del os.environ['NBASR_CELL_STAMPS']
s = stamps.cpu().numpy()
hw, xcc = s[:, 0], s[:, 1] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
t = s[:, 2:13].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0          # us
names = ['start', 'loaded', 'sync', 'conv0', 'epi0', 'conv1', 'epi1', 'conv2', 'epi2(stores issued)', 'end']
print(f'block {blk} B={b}: {nwg} workgroups, kernel span {t[:, :10].max():.1f} us')
dur = np.diff(t[:, :10], axis=1)
for i, n in enumerate(names[1:]):
    print(f'  {names[i]:>22s} -> {n:<22s}: mean {dur[:, i].mean():6.2f} us  p10 {np.percentile(dur[:, i], 10):6.2f}  p90 {np.percentile(dur[:, i], 90):6.2f}')
life = t[:, 9] - t[:, 0]
print(f'  workgroup lifetime mean {life.mean():.2f} us, p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}')
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
ids = np.unique(cuid)
print(f'  {len(ids)} distinct CUs; workgroups per CU: min {min((cuid == i).sum() for i in ids)}, max {max((cuid == i).sum() for i in ids)}')
# one CU's timeline
for i in ids[:2]:
    rows = np.where(cuid == i)[0]
    rows = rows[np.argsort(t[rows, 0])]
    print(f'  CU {i}:')
    for r in rows:
        print('     wg %5d: ' % r + ' '.join(f'{v:7.2f}' for v in t[r, :10]))
# concurrency: average number of workgroups in the conv phases / memory phases over time
grid = np.linspace(0, t[:, 9].max(), 400)
inconv = np.zeros_like(grid); inmem = np.zeros_like(grid)
for r in range(nwg):
    for (a_, b_) in ((2, 3), (4, 5), (6, 7)):
        inconv += (grid >= t[r, a_]) & (grid < t[r, b_])
    inmem += ((grid >= t[r, 0]) & (grid < t[r, 1])) | ((grid >= t[r, 7]) & (grid < t[r, 9]))
print('  workgroups in a conv loop (chip-wide) over time:', ' '.join(f'{int(v)}' for v in inconv[::20]))
print('  workgroups in load/epi2+store phase over time:  ', ' '.join(f'{int(v)}' for v in inmem[::20]))
