#!/usr/bin/env python3
"""Copies the reference's TIMIT normalisation statistics (a DATA file: training/timit_train_stats.npz, read by
training/torch/timit.py:78-84) into tests/golden/frontend_fixtures.npz.  Run in the build container, where /root/reference
exists:  python tests/golden/make_frontend_golden.py"""
import pathlib

import numpy as np

src = pathlib.Path('/root/reference/nasbench_asr/training/timit_train_stats.npz')
stats = np.load(src)
out = pathlib.Path(__file__).resolve().parent / 'frontend_fixtures.npz'
np.savez(out, moving_mean=stats['moving_mean'].astype(np.float32), moving_variance=stats['moving_variance'].astype(np.float32))
print(f'wrote {out}: mean[{stats["moving_mean"].shape[0]}], variance[{stats["moving_variance"].shape[0]}]')
