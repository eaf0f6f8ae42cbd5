#!/usr/bin/env python3
"""tools/ubench/dense_tiles.py output (one or more batches) -> nb_asr_amd/dense_tile_table.json: measured us per launch of the image-path dense
convolution for every (row tile, frame tile), keyed by "c_in,c_out,stride,frames_out" and batch.  executor._dense_tile reads it.

usage: python tools/make_dense_tile_table.py gpurun_out/.../dense_tiles.txt > nb_asr_amd/dense_tile_table.json"""
import collections
import json
import re
import sys

SHAPES = {'conv_0': (80, 600, 1), 'conv_1': (600, 800, 1), 'conv_2': (800, 1000, 2), 'conv_3': (1000, 1200, 2)}
FRAMES_OUT = {'conv_0': 1000, 'conv_1': 1000, 'conv_2': 500, 'conv_3': 250}
table = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r'(conv_\d) b=(\d+) rows=\s*(\d+) frames=\s*(\d+) workgroups=\s*\d+:\s*([\d.]+) us', line)
    if not m:
        continue
    name, b, rows, ft, us = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
    cin, cout, s = SHAPES[name]
    key = f'{cin},{cout},{s},{FRAMES_OUT[name]}'
    table.setdefault(key, collections.OrderedDict()).setdefault(str(b), collections.OrderedDict())[f'{rows}x{ft}'] = us
print(json.dumps({'_doc': 'us per launch, image-path fp16 dense conv with the statistics by-product, one MI355X (tools/ubench/dense_tiles.py); '
                          'key: c_in,c_out,stride,frames_out -> batch -> "rows x frames" tile', 'table': table}, indent=1))
