"""NAS-Bench-ASR search space: op table, enumeration, hashing, arch_vec -> op names.

Host-side counterpart of the reference's ``nasbench_asr/search_space.py`` (SURVEY.md section 8
rows a15 and a17):

* ``all_ops`` order / op ids                      -- reference ``search_space.py:6``
* ``get_search_space``                            -- ``search_space.py:11-18``
* ``get_all_architectures`` (first dim fastest)   -- ``search_space.py:32-47``
* ``get_random_architectures``                    -- ``search_space.py:50-64``
* ``get_model_hash``                              -- ``search_space.py:21-29``
* ``arch_vec_to_names``                           -- ``search_space.py:77-93``

An architecture vector is ``[[op, s0], [op, s0, s1], [op, s0, s1, s2]]``: per cell node the id of
its main operation followed by one 0/1 flag per possible skip input (the cell input and every
earlier node).
"""
import itertools
import random

from .utils import flatten, copy_structure

all_ops = ['linear', 'conv5', 'conv5d2', 'conv7', 'conv7d2', 'zero']
ops_no_zero = all_ops[:-1]
default_nodes = 3


def get_search_space(ops=None, nodes=None):
    """Number of choices per position, nested like an arch_vec: ``[[6,2],[6,2,2],[6,2,2,2]]``."""
    n_ops = len(all_ops if ops is None else ops)
    n_nodes = default_nodes if nodes is None else nodes
    return [[n_ops] + [2] * (k + 1) for k in range(n_nodes)]


def get_all_architectures(ops=None, nodes=None):
    """Yield every arch_vec of the space; the FIRST flat position varies fastest.

    (The reference enumerates with an odometer whose least-significant digit is position 0;
    ``itertools.product`` varies the last factor fastest, so the factors are reversed.)
    """
    shape = get_search_space(ops, nodes)
    radices = flatten(shape)
    for digits in itertools.product(*(range(r) for r in reversed(radices))):
        yield copy_structure(list(reversed(digits)), shape)


def get_random_architectures(num, ops=None, nodes=None, seed=None):
    """``num`` uniformly random arch_vecs (python ``random``; optional seed)."""
    if seed is not None:
        random.seed(seed)
    shape = get_search_space(ops, nodes)
    radices = flatten(shape)
    return [copy_structure([random.randrange(r) for r in radices], shape) for _ in range(num)]


def get_model_hash(arch_vec, ops=None, minimize=True):
    """Isomorphism-invariant MD5 fingerprint of the cell graph described by ``arch_vec``."""
    from . import graph_utils
    graph, _ = graph_utils.get_model_graph(arch_vec, ops=ops, minimize=minimize)
    return graph_utils.graph_hash(graph)


def get_unique_architectures(ops=None, nodes=None):
    """First enumerated representative of every distinct model hash, in enumeration order.

    This is the work list of BASELINE.json config 5 (8 242 entries for the default space).
    """
    seen = {}
    for arch in get_all_architectures(ops, nodes):
        h = get_model_hash(arch, ops=ops)
        if h not in seen:
            seen[h] = arch
    return seen


def get_archs_with_zero():
    """Unique architectures that use the ``zero`` op, ordered by hash."""
    zero_id = all_ops.index('zero')
    by_hash = {}
    for arch in get_all_architectures(all_ops, default_nodes):
        if zero_id in flatten(arch):
            by_hash[get_model_hash(arch)] = arch
    return [by_hash[h] for h in sorted(by_hash)]


def arch_vec_to_names(arch_vec, ops=None):
    """Replace each node's op id by its name; skip flags are kept as they are.

    NB: like the reference (``search_space.py:93``) the canonical ``all_ops`` table is what gets
    indexed, even when a custom ``ops`` list is passed.
    """
    return [[all_ops[node[0]]] + list(node[1:]) for node in arch_vec]
