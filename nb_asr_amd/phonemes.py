"""TIMIT phoneme sets and their folding (reference training/torch/encoder.py ``PhonemeEncoder``).

The table is the standard 61 -> 48 -> 39 folding of Lee & Hon (1989): training labels use the 48-phoneme set (plus the CTC
blank = 0, classes numbered from 1 in sorted order, encoder.py:38-49), scoring folds to 39.  ``''`` = dropped ('q').
"""
import torch

# (61-set, 48-set, 39-set)
FOLDING = tuple(tuple(row.split('/')) for row in (
    'aa/aa/aa ae/ae/ae ah/ah/ah ao/ao/aa aw/aw/aw ax/ax/ah ax-h/ax/ah axr/er/er ay/ay/ay b/b/b bcl/vcl/sil ch/ch/ch d/d/d '
    'dcl/vcl/sil dh/dh/dh dx/dx/dx eh/eh/eh el/el/l em/m/m en/en/n eng/ng/ng epi/epi/sil er/er/er ey/ey/ey f/f/f g/g/g '
    'gcl/vcl/sil h#/sil/sil hh/hh/hh hv/hh/hh ih/ih/ih ix/ix/ih iy/iy/iy jh/jh/jh k/k/k kcl/cl/sil l/l/l m/m/m n/n/n ng/ng/ng '
    'nx/n/n ow/ow/ow oy/oy/oy p/p/p pau/sil/sil pcl/cl/sil q// r/r/r s/s/s sh/sh/sh t/t/t tcl/cl/sil th/th/th uh/uh/uh uw/uw/uw '
    'ux/uw/uw v/v/v w/w/w y/y/y z/z/z zh/zh/sh').split())
ENCODINGS = (61, 48, 39)


def vocab(num_classes, inc_blank=False):
    """Sorted class names of one label set (``PhonemeEncoder.get_vocab``, encoder.py:51-56)."""
    col = ENCODINGS.index(num_classes)
    names = sorted({row[col] for row in FOLDING if row[col]})
    return (['_'] if inc_blank else []) + names


def index_mapping(src_classes, dst_classes):
    """{source label: destination label} with 0 = blank (encoder.py:41-49)."""
    src, dst = ENCODINGS.index(src_classes), ENCODINGS.index(dst_classes)
    if dst <= src:
        raise ValueError(f'cannot fold {src_classes} classes to {dst_classes}')
    fold = {row[src]: row[dst] for row in FOLDING}
    names = vocab(dst_classes)
    mapping = {0: 0}
    for i, ph in enumerate(vocab(src_classes)):
        mapping[i + 1] = names.index(fold[ph]) + 1 if fold[ph] else 0
    return mapping


def fold_table(src_classes=48, dst_classes=39, sequential=True):
    """Label lookup table (int32 tensor, ``src_classes + 1`` entries) that folds one label set into a smaller one.

    ``sequential=True`` reproduces the reference exactly: ``PhonemeEncoder.fold_encoded`` (encoder.py:71-73) relabels the
    tensor in place, one source label after the other in increasing order, so a label that was just moved UP is moved again
    when the loop reaches its new index -- 'cl' and 'epi' end as 'ng' instead of 'sil', 'el' as 'hh' instead of 'l', 'en'
    as 'ih' instead of 'n' (tests/golden/decode_fixtures.npz holds the reference's own output).  ``sequential=False`` is the
    intended one-step folding."""
    mapping = index_mapping(src_classes, dst_classes)
    table = list(range(len(mapping)))
    if sequential:
        for old, new in mapping.items():
            table = [new if v == old else v for v in table]
    else:
        table = [mapping[v] for v in table]
    return torch.tensor(table, dtype=torch.int32)
