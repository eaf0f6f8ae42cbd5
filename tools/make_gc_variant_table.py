#!/usr/bin/env python3
"""Builds nb_asr_amd/gc_variant_table.json -- which variant of the fp32 node kernel the executor launches per shape class -- from
the A/B measurements of tools/ubench/ab_gc_variants.py committed under profiles/r03_gc_variants/.

    python tools/make_gc_variant_table.py

Key "k,d,cg,flavour,size[,stats]": kernel taps, dilation, channels per group, flavour = lnx+skip | lnx (LayerNorm on load of the main
input, with / without a skip input) | skip (at least one skip input) | plain, size = small (under 4 default-kernel waves per SIMD:
measured at 8 utterances) | large (measured at 64).  Value: NBASR_GC_* bits (0 default, 4 output split, 8 pipelined loads, 12 both,
16 LDS ring); with ",stats" the choice among the variants that have a statistics epilogue (measured in that
flavour).  A variant must beat the default by 2 % to be chosen.
"""
import glob
import json
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
table = {}
for f in sorted(glob.glob(str(ROOT / 'profiles/r03_gc_variants/*.jsonl'))):
    for line in open(f):
        if not line.startswith('{'):
            continue
        d = json.loads(line)
        if d['batch'] not in (8, 64):
            continue
        size = 'small' if d['batch'] == 8 else 'large'
        t = {int(k[1:]): v for k, v in d.items() if k[0] == 'v' and k[1:].isdigit()}
        best = min(t, key=t.get)
        choice = 0 if t[0] <= 1.02 * t[best] else best
        base = f"{d['k']},{d['d']},{d['C'] // 100}"
        if d['flavour'] in ('stats0', 'stats'):
            table[f"{base},{'plain' if d['flavour'] == 'stats0' else 'skip'},{size},stats"] = choice
        else:
            table[f"{base},{d['flavour']},{size}"] = choice
out = ROOT / 'nb_asr_amd/gc_variant_table.json'
out.write_text(json.dumps(dict(sorted(table.items())), indent=0) + '\n')
print(f'{len(table)} entries -> {out}')
