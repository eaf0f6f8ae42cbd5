#!/usr/bin/env python3
"""Builds nb_asr_amd/gc_variant_table.json -- which variant of the fp32 node kernel the executor launches per shape class -- from
the A/B measurements of tools/ubench/ab_gc_osplit.py committed under profiles/r02_gc_variants2/.

    python tools/make_gc_variant_table.py

Key "k,d,cg,flavour,size[,stats]": kernel taps, dilation, channels per group, flavour = lnx (LayerNorm on load of the main input)
| skip (at least one skip input) | plain, size = small (default kernel under 4 waves per SIMD: measured at 8 utterances) | large
(measured at 64).  Value: NBASR_GC_* bits (0 default, 4 output split, 8 pipelined loads, 12 both); with ",stats" the choice among
the variants that have a statistics epilogue (0, 8).  A variant must beat the default by 2 % to be chosen.
"""
import glob
import json
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
NAMES = {0: 'default_us', 4: 'osplit_us', 8: 'pipe_us', 12: 'pipe_osplit_us'}
table = {}
for f in sorted(glob.glob(str(ROOT / 'profiles/r02_gc_variants2/*.jsonl'))):
    for line in open(f):
        if not line.startswith('{'):
            continue
        d = json.loads(line)
        if d['batch'] not in (8, 64):
            continue
        size = 'small' if d['batch'] == 8 else 'large'
        flavour = 'lnx' if d['flavour'] == 'lnx' else ('skip' if d['skips'] else 'plain')
        key = f"{d.get('k', 5)},{d.get('d', 1)},{d['C'] // 100},{flavour},{size}"
        t = {v: d[n] for v, n in NAMES.items()}
        best = min(t, key=t.get)
        table[key] = 0 if t[0] <= 1.02 * t[best] else best
        table[key + ',stats'] = 8 if t[8] < 0.98 * t[0] else 0
out = ROOT / 'nb_asr_amd/gc_variant_table.json'
out.write_text(json.dumps(dict(sorted(table.items())), indent=0) + '\n')
print(f'{len(table)} entries -> {out}')
