"""Deterministic, platform-independent parameter fills keyed by ``state_dict`` key.

``get_model`` initialises like the reference (Xavier-uniform through the torch RNG,
``model/torch/__init__.py:13-29``).  Tests, golden fixtures and ``bench.py`` need weights that are
bit-identical on every box and torch version, so they overwrite the parameters with the
counter-based generator of ``utils.keyed_uniform`` -- on this package's model AND on the reference
model when fixtures are generated (same keys, same shapes => same weights).

modes
  'xavier' : the reference's init distribution -- weights U(+-sqrt(6/(fan_in+fan_out))), biases 0,
             LayerNorm 1/0.  With no skip connections activations decay ~30x per cell.
  'lively' : He-uniform conv / op-linear weights (U(+-sqrt(6/fan_in))), small non-zero biases,
             LayerNorm gamma 1 +- 0.2, beta +- 0.1: every code path (bias, gamma, beta) is exercised
             and activations stay O(1) at any depth, so end-to-end tolerances are meaningful.
"""
import math

import torch

from .utils import keyed_uniform


def _fans(shape):
    receptive = 1
    for s in shape[2:]:
        receptive *= s
    return shape[1] * receptive, shape[0] * receptive


def keyed_values(key, shape, seed=1235, mode='xavier'):
    """The values ``keyed_fill_`` gives the parameter ``key`` of ``shape``: a float32 CPU tensor."""
    if mode not in ('xavier', 'lively'):
        raise ValueError(f'unknown mode {mode!r}')
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in, fan_out = _fans(shape)
        he = mode == 'lively' and ('.conv.' in key or '.op.linear.' in key)
        bound = math.sqrt(6.0 / fan_in) if he else math.sqrt(6.0 / (fan_in + fan_out))
        return torch.from_numpy(keyed_uniform(key, seed, shape, -bound, bound))
    if key.endswith('weight'):                            # LayerNorm gamma
        return torch.from_numpy(keyed_uniform(key, seed, shape, 0.8, 1.2)) if mode == 'lively' else torch.ones(shape)
    # biases and LayerNorm beta
    return torch.from_numpy(keyed_uniform(key, seed, shape, -0.1, 0.1)) if mode == 'lively' else torch.zeros(shape)


def keyed_fill_(model, seed=1235, mode='xavier'):
    """Overwrite every parameter of ``model`` in place; returns ``model``."""
    if mode not in ('xavier', 'lively'):
        raise ValueError(f'unknown mode {mode!r}')
    with torch.no_grad():
        for key, p in model.state_dict().items():
            p.copy_(keyed_values(key, tuple(p.shape), seed, mode).to(p.device))
    return model


def keyed_input(batch, frames, seed=0, features=80):
    """Synthetic normalised log-mel batch x ~ N(0, 1), shape (batch, features, frames), float32 CPU."""
    from .utils import keyed_normal
    return torch.from_numpy(keyed_normal(f'input:{batch}x{features}x{frames}', seed, (batch, features, frames)))
