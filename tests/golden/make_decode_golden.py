#!/usr/bin/env python3
"""Golden vectors for the phoneme folding of the reference's validation decode.

Runs the reference's own PhonemeEncoder (training/torch/encoder.py, pure Python; loaded by file path because the package's
__init__ imports the trainer and with it ctcdecode, which is not installed) and stores inputs/outputs only:
  rows61/48/39   the three columns of the reference's DATA file training/timit_folding.txt
  vocab48/39     get_vocab(num_classes=...)
  map48to39      idx_mappings[1][2] as an array (index = 48-set label, 0 = blank)
  seq_in/seq_out fold_encoded(seq_in, 39) on all labels 0..48 and on random label matrices (the in-place sequential relabelling)
Run in the build container, where /root/reference exists:  python tests/golden/make_decode_golden.py"""
import importlib.util
import pathlib

import numpy as np
import torch

REF = pathlib.Path('/root/reference/nasbench_asr/training')
spec = importlib.util.spec_from_file_location('ref_encoder', REF / 'torch' / 'encoder.py')
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

enc = mod.PhonemeEncoder(48)
rows = [mod.split_with_pad(line, '\t', 3) for line in (REF / 'timit_folding.txt').read_text().strip().split('\n')]
mapping = enc.idx_mappings[1][2]
gen = torch.Generator().manual_seed(39)
seq_in = torch.cat([torch.arange(49).view(1, 49), torch.randint(0, 49, (7, 49), generator=gen)]).to(torch.int32)
seq_out = enc.fold_encoded(seq_in.clone(), 39)
same = enc.fold_encoded(seq_in.clone(), 48)                     # num_classes >= own: returned unchanged (encoder.py:65-66)
assert torch.equal(same, seq_in)
out = pathlib.Path(__file__).resolve().parent / 'decode_fixtures.npz'
np.savez(out,
         rows61=np.array([r[0] for r in rows]), rows48=np.array([r[1] for r in rows]), rows39=np.array([r[2] for r in rows]),
         vocab48=np.array(enc.get_vocab()), vocab39=np.array(enc.get_vocab(num_classes=39)),
         vocab48_blank=np.array(enc.get_vocab(inc_blank=True)),
         map48to39=np.array([mapping[i] for i in range(49)], dtype=np.int32),
         seq_in=seq_in.numpy(), seq_out=seq_out.numpy())
print(f'wrote {out}: {len(rows)} folding rows, {seq_in.shape} label matrix')
