"""Recipes shared by the golden generator's consumers: how each fixture's inputs are rebuilt.

Mirrors the construction in tests/golden/make_golden.py (which ran the reference); here the same
keyed inputs / parameters are rebuilt WITHOUT the reference so that the oracle (CPU) and the HIP
kernels (GPU) can be compared with the stored expected outputs.
"""
import numpy as np
import torch

from nb_asr_amd.utils import keyed_uniform, keyed_normal

ARCH_A = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]
ARCH_D = [[3, 1], [4, 1, 1], [2, 1, 1, 1]]
ARCH_M = [[0, 1], [5, 1, 0], [2, 0, 1, 1]]
ARCHS = {'A': ARCH_A, 'D': ARCH_D, 'M': ARCH_M}

MODEL_CASES = [
    # tag, arch, use_rnn, init mode, batch, frames
    ('A_xavier_b1_t500', ARCH_A, True, 'xavier', 1, 500),
    ('A_lively_b1_t500', ARCH_A, True, 'lively', 1, 500),
    ('A_lively_b2_t67', ARCH_A, True, 'lively', 2, 67),
    ('D_xavier_b1_t200', ARCH_D, True, 'xavier', 1, 200),
    ('M_lively_b2_t40_nornn', ARCH_M, False, 'lively', 2, 40),
    ('M_lively_b1_t90', ARCH_M, True, 'lively', 1, 90),
]
N_SAMPLES = 16

GCONV_CASES = [(cg, k, d) for cg in (6, 8, 10, 12) for k, d in ((5, 1), (5, 2), (7, 1), (7, 2))]
GCONV100_CASES = [('conv5', 600, 5, 1), ('conv5d2', 800, 5, 2), ('conv7', 1000, 7, 1), ('conv7d2', 1200, 7, 2)]
DENSE_CASES = [(24, 40, 37, 1, 2), (24, 40, 37, 2, 2), (24, 40, 64, 2, 1), (80, 600, 50, 1, 1), (136, 200, 131, 2, 1),
               (136, 200, 300, 1, 1)]
LINEAR_CASES = [(24, 37, 2), (600, 16, 1), (200, 140, 1)]
LAYERNORM_CASES = [(600, 19), (1200, 7), (24, 37)]
LSTM_CASES = [(16, 8, 9, 3), (40, 20, 33, 18)]
NODE_OPS = ('conv5', 'conv7d2', 'linear', 'zero')


def keyed_params(shapes, tag, seed=7, bias_scale=0.2):
    """{key: tensor} exactly as make_golden.fill_module_ filled the reference module."""
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        if len(shape) >= 2:
            bound = (6.0 / int(np.prod(shape[1:]))) ** 0.5
            vals = keyed_uniform(f'{tag}/{key}', seed, shape, -bound, bound)
        elif key.endswith('weight'):
            vals = keyed_uniform(f'{tag}/{key}', seed, shape, 0.7, 1.3)
        else:
            vals = keyed_uniform(f'{tag}/{key}', seed, shape, -bias_scale, bias_scale)
        out[key] = torch.from_numpy(vals)
    return out


def keyed_x(tag, shape, scale=1.0):
    return torch.from_numpy(keyed_normal(tag + '/x', 3, shape)) * scale


def sample_indices(tag, idx, numel):
    u = keyed_uniform(f'{tag}/layer{idx}/samples', 11, (N_SAMPLES,), 0.0, 1.0).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def node_shapes(op_name, c):
    if op_name == 'linear':
        return {'op.linear.weight': (c, c), 'op.linear.bias': (c,)}
    if op_name == 'zero':
        return {}
    k = {'conv5': 5, 'conv5d2': 5, 'conv7': 7, 'conv7d2': 7}[op_name]
    return {'op.conv.weight': (c, c // 100, k), 'op.conv.bias': (c,)}


def cell_shapes(arch, c, use_norm):
    from oracle import asr_oracle as oracle
    shapes = {}
    for j, (op_name, *_f) in enumerate(oracle.arch_names(arch)):
        for key, shp in node_shapes(op_name, c).items():
            shapes[f'nodes.{j}.{key}'] = shp
    if use_norm:
        shapes['norm_layer.weight'] = (c,)
        shapes['norm_layer.bias'] = (c,)
    return shapes


def worst_ratio(got, want, rtol, atol):
    got = torch.as_tensor(got).double().cpu()
    want = torch.as_tensor(want).double().cpu()
    err = (got - want).abs()
    return float((err / (atol + rtol * want.abs())).max()) if err.numel() else 0.0
