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


def register_with(nasbench_asr=None, default=False):
    """Make this package the ``'hip'`` model backend of an imported ``nasbench_asr`` -- the reference needs NO source change.

    The reference resolves a backend name through ``nasbench_asr.model._backends`` (``utils.BackendsAccessor``,
    reference ``utils.py:115-165``): ``get_backend(name)`` first looks the name up in its ``backends`` dict and only then
    validates / imports a sub-package (``utils.py:150-153``).  Seeding that dict is therefore all a registration takes:

        import nasbench_asr, nb_asr_amd
        nb_asr_amd.register_with(nasbench_asr)
        model = nasbench_asr.get_model(arch_vec, use_rnn=True, dropout_rate=0.0, gpu=0, backend='hip')   # model/__init__.py:19-20
        nasbench_asr.model.print_model_summary(model)           # dispatches on model.backend == 'hip' (model/__init__.py:23-24)

    ``default=True`` also makes it what ``get_model(...)`` without ``backend=`` returns (the accessor's ``None`` entry, which
    the reference itself fills on first use, ``utils.py:163-164``).  Returns the backend name."""
    import sys
    if nasbench_asr is None:
        import nasbench_asr
    accessor = nasbench_asr.model._backends
    this = sys.modules[__name__]
    accessor.backends[_BACKEND] = this
    if _BACKEND not in accessor.available_backends:
        accessor.available_backends.append(_BACKEND)
    if default:
        accessor.backends[None] = this
    return _BACKEND


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
