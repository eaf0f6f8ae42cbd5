"""Host-side logic vs known answers produced by the reference (tests/golden/host_known_answers.json).

Covers SURVEY.md section 8 rows a1 (pad rule), a9 (T -> T'), a15 (names), a17 (enumeration + graph hash)
and the nested-list helpers.  No GPU needed.
"""
import hashlib
import json

import pytest

import nb_asr_amd as nb
from nb_asr_amd import graph_utils, hip, search_space, utils
from nb_asr_amd.weights import keyed_fill_, keyed_input


def _sha(text):
    return hashlib.sha256(text.encode('utf-8')).hexdigest()


def test_readme_hash(known):
    assert search_space.get_model_hash(known['readme_hash']['arch']) == known['readme_hash']['hash']
    assert known['readme_hash']['hash'] == '36855332a5778e0df5114305bc3ce238'       # reference README.md:61


def test_search_space_shape_and_ops(known):
    assert search_space.get_search_space() == known['search_space'] == [[6, 2], [6, 2, 2], [6, 2, 2, 2]]
    assert search_space.all_ops == known['all_ops']
    assert search_space.get_search_space(ops=['a', 'b'], nodes=2) == [[2, 2], [2, 2, 2]]


def test_enumeration_order_and_hashes(known):
    archs = list(search_space.get_all_architectures())
    assert len(archs) == known['n_points'] == 13824
    assert archs[:8] == known['first_archs']
    assert archs[-1] == known['last_arch']
    assert _sha(json.dumps(archs)) == known['enumeration_sha256']
    hashes = [search_space.get_model_hash(a) for a in archs]
    assert _sha(''.join(hashes)) == known['hashes_sha256']                      # all 13 824, in order
    assert len(set(hashes)) == known['n_unique'] == 8242
    no_zero = {h for a, h in zip(archs, hashes) if 5 not in utils.flatten(a)}
    assert len(no_zero) == known['n_unique_no_zero'] == 8000
    for arch_json, h in known['sample_hashes'].items():
        assert search_space.get_model_hash(json.loads(arch_json)) == h


def test_unique_architectures_are_first_representatives(known):
    uniq = search_space.get_unique_architectures()
    assert len(uniq) == known['n_unique']
    seen = set()
    for arch in search_space.get_all_architectures():
        h = search_space.get_model_hash(arch)
        if h not in seen:
            seen.add(h)
            assert uniq[h] == arch
        if len(seen) > 300:
            break


def test_hash_without_minimisation_and_graphs(known):
    for arch_json, h in known['hash_no_minimize'].items():
        assert search_space.get_model_hash(json.loads(arch_json), minimize=False) == h
    for arch_json, g in known['graphs'].items():
        (adj, labels), original = graph_utils.get_model_graph(json.loads(arch_json))
        assert adj.astype(int).tolist() == g['adjacency']
        assert labels == g['labels']
        assert original is not None and original[0].shape == (5, 5)


def test_hash_is_isomorphism_invariant():
    # the two skip sources of the last node are symmetric when both earlier nodes are `zero`-free twins
    a = [[1, 0], [1, 1, 0], [5, 0, 0, 0]]
    b = [[1, 0], [1, 1, 0], [5, 1, 1, 1]]
    # a zero op cuts everything behind it: skip flags of a node feeding only a dead end do not matter
    assert search_space.get_model_hash([[5, 0], [5, 0, 0], [5, 0, 0, 0]]) == search_space.get_model_hash([[5, 1], [5, 0, 1], [5, 0, 0, 0]])
    assert search_space.get_model_hash(a) != search_space.get_model_hash(b)


def test_arch_vec_to_names(known):
    for arch_json, names in known['names'].items():
        assert search_space.arch_vec_to_names(json.loads(arch_json)) == names
    # like the reference, the canonical op table is used even when `ops` is given (search_space.py:93)
    assert search_space.arch_vec_to_names([[0, 1]], ops=['x', 'y'])[0][0] == 'linear'


def test_random_architectures_are_seeded_and_in_range():
    a = search_space.get_random_architectures(20, seed=3)
    b = search_space.get_random_architectures(20, seed=3)
    assert a == b and len(a) == 20
    shape = search_space.get_search_space()
    for arch in a:
        assert [len(n) for n in arch] == [len(n) for n in shape]
        assert all(0 <= v < r for v, r in zip(utils.flatten(arch), utils.flatten(shape)))


def test_nested_helpers(known):
    seq = [[1, 2], [3, [4, 5]], 6]
    assert utils.flatten(seq) == [1, 2, 3, 4, 5, 6]
    assert utils.copy_structure(utils.flatten(seq), seq) == seq                 # utils.py:80-84 docstring property
    assert utils.copy_structure(range(3), (0, (0, 0))) == (0, (1, 2))
    assert utils.count(x for x in range(7)) == 7
    assert list(utils.get_first_n(iter(range(100)), 3)) == [0, 1, 2]
    for n, text in known['nice_numbers'].items():
        assert utils.make_nice_number(int(n)) == text


def test_pad_rule_through_the_c_abi(known):
    for key, (left, right) in known['pads'].items():
        k, d, s = (int(v) for v in key.split(','))
        assert hip.pad_amounts(k, d, s) == (left, right)
    assert hip.pad_amounts(1, 1, 1) == (0, 0)
    with pytest.raises(hip.HipError):
        hip.pad_amounts(0, 1, 1)


def test_output_frames(known):
    for t, t_out in known['out_frames'].items():
        assert hip.output_frames(int(t)) == t_out
    assert hip.output_frames(0) == 0


def test_keyed_generators_are_platform_stable(known):
    import torch
    m = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
    for mode, digests in known['keyed_fill_digests'].items():
        keyed_fill_(m, seed=1235, mode=mode)
        sd = m.state_dict()
        for key, want in digests.items():
            got = hashlib.sha256(sd[key].detach().cpu().numpy().tobytes()).hexdigest()
            assert got == want, (mode, key)
    x = keyed_input(2, 37, seed=0)
    assert hashlib.sha256(x.numpy().tobytes()).hexdigest() == known['keyed_input_digest']
    assert x.dtype == torch.float32 and tuple(x.shape) == (2, 80, 37)


def test_row_tile_rule_counts_rounds_of_workgroups():
    """Image-path GEMM: 160-row tiles exactly where they mean fewer (rows x rounds of 256 workgroups) than 128-row tiles."""
    import types
    from nb_asr_amd.executor import ForwardPlan
    plan = types.SimpleNamespace(batch=64)
    pick = lambda c, t: ForwardPlan._row_tile(plan, c, t)                      # noqa: E731
    assert [pick(800, 1000), pick(1000, 500), pick(1200, 250)] == [160, 128, 160]     # the benchmark shape: convs 1 and 3
    plan.batch = 32
    assert [pick(800, 1000), pick(1000, 500), pick(1200, 250)] == [160, 128, 160]
    plan.batch = 8
    assert [pick(600, 1000), pick(800, 1000), pick(1000, 500), pick(1200, 250)] == [96, 128, 64, 64]    # under-filled launches: 64- / 96-row tiles
    plan.batch = 16
    assert [pick(600, 1000), pick(800, 1000), pick(1000, 500), pick(1200, 250)] == [160, 128, 128, 96]  # (round 4) conv 3: 13 x 16 = 208 workgroups, one round
    plan.batch = 2
    assert [pick(800, 1000), pick(1000, 500), pick(1200, 250)] == [64, 64, 64]        # a single round whatever the tile: the smallest
    assert ForwardPlan._row_tile(plan, 1200, 250, allow_64=False) == 128               # (the bf16 GEMM has no 64-row instance)
