"""nb_asr_amd -- MI355X (gfx950) native forward pass of the NAS-Bench-ASR acoustic model.

Drop-in for the model-construction / forward surface of ``nasbench_asr`` (reference
``nasbench_asr/__init__.py:38-40`` -> ``model/__init__.py:19-24`` -> ``model/torch/__init__.py``):

    import nb_asr_amd as nb
    model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0, gpu=0)
    logits = model.eval()(x)            # x: (B, 80, T) float32 on the HIP device -> (B, T', 49)

Training, data loading and the dataset query API of the reference are out of scope.
"""
import torch
import torch.nn as _nn

from . import search_space
from . import graph_utils
from . import utils
from . import ctc
from . import frontend
from .search_space import (all_ops, get_search_space, get_all_architectures, get_random_architectures,
                           get_model_hash, arch_vec_to_names)

__version__ = '0.1.0'

_BACKEND = 'hip'


def get_available_backends():
    return [_BACKEND]


def get_backend_name(backend=None):
    if backend not in (None, _BACKEND):
        raise ValueError(f'Unknown backend: {backend}')
    return _BACKEND


def set_default_backend(backend):
    return get_backend_name(backend)


def get_model(arch_vec, use_rnn, dropout_rate, gpu=None, backend=None):
    """Build the model for ``arch_vec`` with the reference's initialisation.

    Same positional signature as the reference backend entry (``model/torch/__init__.py:7``):
    Xavier-uniform weights and zero biases for every Linear / Conv1d / LSTM, LayerNorm left at 1/0,
    module returned in training mode, moved to ``cuda:{gpu}`` (the HIP device) when ``gpu`` is given.
    """
    get_backend_name(backend)
    from .model import ASRModel
    model = ASRModel(arch_vec_to_names(arch_vec), use_rnn=use_rnn, dropout_rate=dropout_rate)

    for m in model.modules():
        if isinstance(m, (_nn.Linear, _nn.Conv1d)):
            _nn.init.xavier_uniform_(m.weight)
            _nn.init.zeros_(m.bias)
        elif isinstance(m, _nn.LSTM):
            for name, p in m.named_parameters():
                if name.startswith('weight'):
                    _nn.init.xavier_uniform_(p)
                else:
                    _nn.init.zeros_(p)
    if gpu is not None:
        model.to(device=f'cuda:{gpu}')
    return model


make_model = get_model       # the name BASELINE.json uses for the same entry point


def print_model_summary(model):
    """Module tree with per-child parameter counts (reference ``model/torch/__init__.py:38-47``)."""
    print(model)
    print('======================')

    def walk(module, depth):
        for name, child in module.named_children():
            print('  ' * depth + type(child).__name__, ' ', name, ' ', sum(p.numel() for p in child.parameters()))
            walk(child, depth + 1)

    walk(model.model, 0)
    print('======================')
    print('Trainable parameters:', utils.make_nice_number(sum(p.numel() for p in model.parameters())))
