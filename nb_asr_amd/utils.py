"""Small host-side helpers used by the search-space and model code.

Counterparts of the nested-list helpers in the reference's ``nasbench_asr/utils.py:63-111``
(``recursive_iter / flatten / copy_structure / count / get_first_n``) and of
``make_nice_number`` (``utils.py:168-175``).  Written from the documented behaviour
(``seq == copy_structure(flatten(seq), seq)``), not from the reference text.

Also holds the build-owned *keyed* weight generator used by tests and by ``bench.py`` so
that identical weights can be produced on any box without depending on the torch RNG stream
(golden-vector design (ii)).
"""
import collections.abc as _abc
import hashlib as _hashlib

import numpy as _np


# ----------------------------------------------------------------------------------------
# nested sequences
# ----------------------------------------------------------------------------------------
def _is_seq(obj):
    return isinstance(obj, _abc.Sequence) and not isinstance(obj, (str, bytes))


def recursive_iter(seq):
    """Depth-first iteration over the leaves of an arbitrarily nested sequence."""
    stack = [iter([seq])]
    while stack:
        try:
            item = next(stack[-1])
        except StopIteration:
            stack.pop()
            continue
        if _is_seq(item):
            stack.append(iter(item))
        else:
            yield item


def flatten(seq):
    """All leaves of ``seq`` as one flat list."""
    return list(recursive_iter(seq))


def copy_structure(data, shape):
    """Pour the flat values of ``data`` into containers nested like ``shape``."""
    leaves = recursive_iter(data)

    def build(template):
        if _is_seq(template):
            return type(template)(build(t) for t in template)
        return next(leaves)

    return build(shape)


def count(seq):
    """Number of items produced by an iterable (consumes it)."""
    n = 0
    for _ in seq:
        n += 1
    return n


def get_first_n(seq, n):
    """Lazily yield the first ``n`` items of an iterable."""
    it = iter(seq)
    for _ in range(n):
        yield next(it)


def make_nice_number(num):
    """``26341349 -> '26,341,349'`` (thousands separators, as the reference's summary prints)."""
    digits = str(num)
    head = len(digits) % 3 or 3
    parts = [digits[:head]] + [digits[i:i + 3] for i in range(head, len(digits), 3)]
    return ','.join(parts)


# ----------------------------------------------------------------------------------------
# keyed, counter-based uniform generator (splitmix64)
# ----------------------------------------------------------------------------------------
_GOLDEN = _np.uint64(0x9E3779B97F4A7C15)
_M1 = _np.uint64(0xBF58476D1CE4E5B9)
_M2 = _np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    x = (x + _GOLDEN)
    z = x
    z = (z ^ (z >> _np.uint64(30))) * _M1
    z = (z ^ (z >> _np.uint64(27))) * _M2
    return z ^ (z >> _np.uint64(31))


def key_to_u64(key, seed):
    """Stable 64-bit stream id for (tensor name, seed)."""
    h = _hashlib.sha256(f'{seed}:{key}'.encode('utf-8')).digest()
    return int.from_bytes(h[:8], 'little')


def keyed_uniform(key, seed, shape, low=-1.0, high=1.0):
    """Deterministic float32 array of ``shape`` with values uniform in [low, high).

    Element ``i`` depends only on (key, seed, i): value = splitmix64(stream + i) >> 40 scaled
    by 2**-24, so the result is identical on every platform and numpy/torch version.
    """
    n = int(_np.prod(shape)) if len(shape) else 1
    with _np.errstate(over='ignore'):
        ctr = _np.arange(n, dtype=_np.uint64) + _np.uint64(key_to_u64(key, seed))
        bits = _splitmix64(ctr) >> _np.uint64(40)          # 24 random bits
    u = bits.astype(_np.float64) * (1.0 / 16777216.0)       # [0, 1)
    out = (low + (high - low) * u).astype(_np.float32)
    return out.reshape(shape)


def keyed_normal(key, seed, shape):
    """Deterministic float32 standard-normal array (Box-Muller over two keyed uniforms)."""
    n = int(_np.prod(shape)) if len(shape) else 1
    u1 = keyed_uniform(key + '#u1', seed, (n,), 0.0, 1.0).astype(_np.float64)
    u2 = keyed_uniform(key + '#u2', seed, (n,), 0.0, 1.0).astype(_np.float64)
    r = _np.sqrt(-2.0 * _np.log(1.0 - u1))                   # 1-u1 in (0, 1]
    z = r * _np.cos(2.0 * _np.pi * u2)
    return z.astype(_np.float32).reshape(shape)
