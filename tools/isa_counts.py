#!/usr/bin/env python3
"""Static instruction counts per kernel instance from hipcc's assembly listing (VERDICT r3 counted the fused cell this way).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Iinclude -Inb_asr_amd/csrc -x hip nb_asr_amd/csrc/grouped_cell.hip -o /tmp/cell.s
    python tools/isa_counts.py /tmp/cell.s 'grouped_cell_kernel<float'
Counts cover the WHOLE kernel function (every (taps, dilation) body of an instance, prologue, epilogues), not one loop."""
import collections
import re
import subprocess
import sys

path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
cur, stats, meta = None, collections.OrderedDict(), {}
with open(path) as f:
    for line in f:
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            stats[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t[0] in '.;/':
            m2 = re.match(r';\s*(NumVgprs|ScratchSize)[:\s]+(\d+)', t)
            if m2:
                meta.setdefault(cur, {})[m2.group(1)] = int(m2.group(2))
            continue
        op, c = t.split()[0], stats[cur]
        if op.startswith('v_'):
            c['valu'] += 1
            if op == 'v_pk_fma_f32':
                c['pk_fma'] += 1
            elif op in ('v_readlane_b32', 'v_writelane_b32'):
                c['lane'] += 1
            elif op.startswith('v_mov') or op.startswith('v_accvgpr'):
                c['mov'] += 1
        elif op.startswith('s_load'):
            c['s_load'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
names = subprocess.run(['c++filt'] + list(stats), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel instance':66s} {'VALU':>6s} {'pk_fma':>6s} {'fma%':>5s} {'lane r/w':>8s} {'lane%':>5s} {'moves':>6s} {'s_load':>6s} {'SALU':>6s} {'LDS':>5s} {'scratch':>7s} {'VGPR':>5s} {'scr B':>6s}")
for k, d in zip(stats, names):
    if want not in d:
        continue
    c, m = stats[k], meta.get(k, {})
    short = re.sub(r'\(.*', '', d).replace('void nbasr::', '')
    print(f"{short:66s} {c['valu']:6d} {c['pk_fma']:6d} {100 * c['pk_fma'] / max(c['valu'], 1):5.1f} {c['lane']:8d} {100 * c['lane'] / max(c['valu'], 1):5.1f} "
          f"{c['mov']:6d} {c['s_load']:6d} {c['salu']:6d} {c['lds']:5d} {c['scratch']:7d} {m.get('NumVgprs', -1):5d} {m.get('ScratchSize', -1):6d}")
