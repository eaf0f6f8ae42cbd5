"""Cell graph construction and isomorphism-invariant hashing of architectures.

Counterpart of the numpy path of the reference's ``nasbench_asr/graph_utils.py``
(``get_model_graph_np`` 17-76, ``graph_hash_np`` 145-180).  Only
used for de-duplicating the 13 824-point search space (config 5) and for dataset keys -- the
forward pass never touches it.

Graph model: vertex 0 is the cell input, vertices 1..n the node operations, vertex n+1 the
output.  Consecutive vertices are chained; the skip flags ``s_0..s_k`` of node ``k`` add edges
``i -> k+2`` (a skip is summed into the node's *output*, i.e. it feeds whatever consumes that
output).  With ``minimize`` every ``zero`` vertex loses its edges and vertices that are not on
some input->output path are dropped.

The fingerprint is the NAS-Bench-101 style iterated neighbourhood hash: every vertex starts
from md5(str((out_degree, in_degree, label))) -- degrees printed as python floats, labels as
ints (-1 input, -2 output, op index otherwise) -- and is mixed ``|V|`` times with the sorted
hashes of its in- and out-neighbours; the result is md5 of the printed sorted list.
"""
import copy
import hashlib

import numpy as np


def _md5(text):
    return hashlib.md5(text.encode('utf-8')).hexdigest()


def _reachable(adj, start, transpose):
    """Vertices reachable from ``start`` following edges (or reversed edges)."""
    n = len(adj)
    seen = [False] * n
    seen[start] = True
    todo = [start]
    while todo:
        v = todo.pop()
        for w in range(n):
            if seen[w]:
                continue
            linked = adj[w][v] if transpose else adj[v][w]
            if linked:
                seen[w] = True
                todo.append(w)
    return seen


def get_model_graph(arch_vec, ops=None, minimize=True, keep_dims=False):
    """Return ``((adjacency, labels), original)``.

    ``adjacency`` is a float numpy matrix (``adjacency[i, j] == 1`` for an edge i->j), ``labels``
    the per-vertex names.  ``original`` is the un-minimised pair (``None`` if ``minimize`` is
    false).
    """
    if ops is None:
        from . import search_space
        ops = search_space.all_ops
    n = len(arch_vec)
    size = n + 2
    adj = [[0] * size for _ in range(size)]
    labels = ['input'] + [ops[node[0]] for node in arch_vec] + ['output']
    for v in range(n + 1):
        adj[v][v + 1] = 1
    for k, node in enumerate(arch_vec):
        for src, flag in enumerate(node[1:]):
            if flag:
                adj[src][k + 2] = 1

    original = None
    if minimize:
        original = (np.array(adj, dtype=np.float64).reshape(size, size), list(labels))
        for v in range(size):
            if labels[v] == 'zero':
                for w in range(size):
                    adj[v][w] = 0
                    adj[w][v] = 0
        fwd = _reachable(adj, 0, transpose=False)
        bwd = _reachable(adj, size - 1, transpose=True)
        alive = [f and b for f, b in zip(fwd, bwd)]
        if not all(alive):
            if keep_dims:
                for v in range(size):
                    if not alive[v]:
                        labels[v] = None
                        for w in range(size):
                            adj[v][w] = 0
                            adj[w][v] = 0
            else:
                keep = [v for v in range(size) if alive[v]]
                adj = [[adj[r][c] for c in keep] for r in keep]
                labels = [labels[v] for v in keep]

    m = len(labels)
    return (np.array(adj, dtype=np.float64).reshape(m, m), labels), original


def graph_hash(graph):
    """MD5 fingerprint of ``(adjacency, labels)``, invariant to vertex relabelling."""
    from . import search_space
    adj, names = graph
    adj = np.asarray(adj)
    n = adj.shape[0]
    if names:
        labels = [-1] + [search_space.all_ops.index(name) for name in names[1:-1]] + [-2]
    else:
        labels = []
    if len(labels) != n:
        raise ValueError(f'label/vertex count mismatch: {labels} vs {n} vertices')

    preds = [[w for w in range(n) if adj[w, v]] for v in range(n)]
    succs = [[w for w in range(n) if adj[v, w]] for v in range(n)]
    # initial colour: "(out_degree, in_degree, label)" with float degrees, exactly as printed by
    # python for a tuple of (float, float, int)
    colours = [_md5(str((float(len(succs[v])), float(len(preds[v])), labels[v]))) for v in range(n)]
    for _ in range(n):
        colours = [
            _md5(''.join(sorted(colours[w] for w in preds[v])) + '|' +
                 ''.join(sorted(colours[w] for w in succs[v])) + '|' + colours[v])
            for v in range(n)
        ]
    return _md5(str(sorted(colours)))


def clone_graph(graph):
    adj, labels = graph
    return copy.copy(adj), list(labels)
