"""The benchmarking-dataset emitter (BASELINE config 5): format round trip, loader checks, work sharding (no GPU)."""
import pathlib
import pickle
import sys

import pytest

from nb_asr_amd import bench_dataset, search_space

ARCH = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]
REF = pathlib.Path('/root/reference')


def _rows():
    return [[search_space.get_model_hash(a), 0.01 * (i + 1)] for i, a in enumerate(([[1, 0], [1, 0, 0], [1, 0, 0, 0]],
                                                                                    [[3, 1], [4, 1, 1], [2, 1, 1, 1]],
                                                                                    [[0, 1], [5, 1, 0], [2, 0, 1, 1]]))]


def test_round_trip_and_header(tmp_path):
    path = tmp_path / bench_dataset.file_name('mi355x-fp32')
    bench_dataset.write_benchmarking_dataset(path, 'mi355x-fp32', _rows(), meta={'batch_size': 32, 'n_gpus': 1})
    assert path.name == 'nb-asr-bench-mi355x-fp32.pickle'
    import json
    meta = json.loads((tmp_path / 'nb-asr-bench-mi355x-fp32.meta.json').read_text())
    assert meta['batch_size'] == 32 and meta['rows'] == 3 and meta['device'] == 'mi355x-fp32'
    with open(path, 'rb') as f:                              # exactly two pickles: header, then rows
        header, data = pickle.load(f), pickle.load(f)
        assert f.read() == b''
    assert header['dataset_type'] == 'benchmarking' and header['device'] == 'mi355x-fp32' and header['version'] >= 1
    # exactly the reference's header keys: its loader compares the headers of all the files it is given (dataset.py:37-42)
    assert sorted(header) == ['columns', 'dataset_type', 'device', 'search_space', 'version']
    assert header['columns'][:2] == ['model_hash', 'latency']
    assert header['search_space'] == {'shape': [[6, 2], [6, 2, 2], [6, 2, 2, 2]], 'ops': search_space.all_ops, 'nodes': 3}
    assert data == _rows()
    hdr, device, db = bench_dataset.read_benchmarking_dataset(path)
    assert device == 'mi355x-fp32' and 'device' not in hdr
    assert db[search_space.get_model_hash(ARCH)] == [0.01]


def test_bad_inputs(tmp_path):
    with pytest.raises(ValueError, match='device name'):
        bench_dataset.make_header('mi355x fp32')
    with pytest.raises(ValueError, match='columns must start'):
        bench_dataset.make_header('gpu', columns=('latency', 'model_hash'))
    with pytest.raises(ValueError, match='does not match columns'):
        bench_dataset.write_benchmarking_dataset(tmp_path / 'x.pickle', 'gpu', [['abc', 0.1, 3]])
    with pytest.raises(ValueError, match='appears twice'):
        bench_dataset.write_benchmarking_dataset(tmp_path / 'x.pickle', 'gpu', [['abc', 0.1], ['abc', 0.2]])
    bad = tmp_path / 'bad.pickle'
    with open(bad, 'wb') as f:
        pickle.dump({'dataset_type': 'training', 'device': 'x', 'columns': []}, f)
    with pytest.raises(ValueError, match='benchmarking'):
        bench_dataset.read_benchmarking_dataset(bad)


def test_weight_bank_builds_models_without_allocating_per_architecture():
    """The sweep's models borrow their parameters from one bank: same tensor objects for the same (key, shape), the
    reference's state_dict keys and shapes, a shared plan pool."""
    import torch
    bank = bench_dataset.WeightBank('cpu')
    a = bank.build([[1, 0], [1, 0, 0], [1, 0, 0, 0]])
    b = bank.build([[0, 1], [5, 1, 0], [3, 0, 1, 1]])
    from oracle import asr_oracle as oracle
    for model, arch in ((a, [[1, 0], [1, 0, 0], [1, 0, 0, 0]]), (b, [[0, 1], [5, 1, 0], [3, 0, 1, 1]])):
        shapes = oracle.parameter_shapes(arch)
        assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        assert not any(p.is_meta for p in model.parameters())
    assert a.model[0].conv.weight is b.model[0].conv.weight and a.model[27].weight_hh_l0 is b.model[27].weight_hh_l0
    assert a._plans is b._plans is bank.pool
    from nb_asr_amd.weights import keyed_values
    assert torch.equal(a.model[0].conv.weight.detach(), keyed_values('model.0.conv.weight', (600, 80, 8), 1235, 'lively'))     # live data, not a constant
    const = bench_dataset.WeightBank('cpu', fill=0.01).build([[1, 0], [1, 0, 0], [1, 0, 0, 0]])                                 # the round-2 fill, kept for A/Bs
    assert float(const.model[1].weight[0]) == 1.0 and abs(float(const.model[0].conv.weight[0, 0, 0]) - 0.01) < 1e-8
    n_unique = len({id(p) for m in (a, b) for p in m.parameters()})
    assert n_unique < sum(1 for m in (a, b) for _ in m.parameters())


def test_summary_by_op_family():
    rows = _rows()
    by_hash = {search_space.get_model_hash(a): a for a in ([[1, 0], [1, 0, 0], [1, 0, 0, 0]], [[3, 1], [4, 1, 1], [2, 1, 1, 1]],
                                                            [[0, 1], [5, 1, 0], [2, 0, 1, 1]])}
    s = bench_dataset.summarize(rows, by_hash)
    assert s['architectures'] == 3 and s['latency_min_s'] == 0.01 and s['latency_max_s'] == 0.03 and s['latency_median_s'] == 0.02
    assert s['median_s_by_op_used']['conv5'] == 0.01 and s['count_by_op_used']['conv5d2'] == 2
    assert set(s['median_s_by_op_used']) == {'conv5', 'conv7', 'conv7d2', 'conv5d2', 'linear', 'zero'}


def test_work_list_shards_cover_every_unique_architecture():
    full = bench_dataset.sweep_work_list(limit=100)
    assert len(full) == 100 and len({h for h, _ in full}) == 100
    assert all(search_space.get_model_hash(a) == h for h, a in full[:10])
    parts = [bench_dataset.sweep_work_list(limit=100, rank=r, world_size=8) for r in range(8)]
    assert sorted(h for p in parts for h, _ in p) == sorted(h for h, _ in full)
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.skipif(not REF.exists(), reason='reference checkout only exists in the build container')
def test_reference_loader_reads_the_file(tmp_path):
    """Where the reference is available, its own BenchmarkingDataset / from_folder must load what we emit."""
    path = tmp_path / bench_dataset.file_name('mi355x-fp32')
    bench_dataset.write_benchmarking_dataset(path, 'mi355x-fp32', _rows())
    sys.path.insert(0, str(REF))
    try:
        from nasbench_asr import dataset as ref_dataset
        ds = ref_dataset.BenchmarkingDataset([str(path)])
        assert ds.devices == ['mi355x-fp32'] and ds.columns[:2] == ['model_hash', 'latency']
        assert ds.latency(ARCH) == [[0.01]]
        assert ds.latency(ARCH, return_dict=True) == {'mi355x-fp32': {'latency': 0.01}}
        assert ds.latency([[5, 0], [5, 0, 0], [5, 0, 0, 0]]) is None
        assert ARCH in ds
        # two sweeps (1 GPU, 8 GPUs) load TOGETHER, and from_folder finds both and ignores the side-car files
        path8 = tmp_path / bench_dataset.file_name('mi355x-fp32-x8')
        bench_dataset.write_benchmarking_dataset(path8, 'mi355x-fp32-x8', [[h, 2 * v] for h, v in _rows()], meta={'n_gpus': 8})
        bench_dataset.write_benchmarking_dataset(path, 'mi355x-fp32', _rows(), meta={'n_gpus': 1})
        both = ref_dataset.BenchmarkingDataset([str(path), str(path8)])
        assert both.devices == ['mi355x-fp32', 'mi355x-fp32-x8']
        assert both.latency(ARCH, return_dict=True) == {'mi355x-fp32': {'latency': 0.01}, 'mi355x-fp32-x8': {'latency': 0.02}}
        # from_folder's file-name pattern (dataset.py:537,544; it also insists on a training file, which a latency sweep
        # does not produce) picks up both pickles and neither side-car
        import re
        pattern = re.compile('nb-asr-bench-[a-zA-Z0-9-]+.pickle')
        assert sorted(f.name for f in tmp_path.iterdir() if pattern.fullmatch(f.name)) == sorted([path.name, path8.name])
        assert sum(f.name.endswith('.meta.json') for f in tmp_path.iterdir()) == 2
    finally:
        sys.path.remove(str(REF))
        for name in [m for m in sys.modules if m.startswith('nasbench_asr')]:
            del sys.modules[name]


def test_committed_full_sweep_covers_the_search_space():
    """profiles/r06_sweep/nb-asr-bench-mi355x-fp32.pickle: BASELINE config 5's artefact, produced on one MI355X by
    tools/latency_sweep.py (8 242 architectures, B=32, T=1000).  One row per unique model hash of the search space, positive
    finite latencies, the reference's header keys only."""
    import math
    path = pathlib.Path(__file__).resolve().parent.parent / 'profiles' / 'r06_sweep' / bench_dataset.file_name('mi355x-fp32')
    header, device, db = bench_dataset.read_benchmarking_dataset(path)
    assert device == 'mi355x-fp32' and sorted(header) == ['columns', 'dataset_type', 'search_space', 'version']
    unique = search_space.get_unique_architectures()
    assert len(db) == len(unique) == 8242 and set(db) == set(unique)
    lat = [v[0] for v in db.values()]
    assert all(math.isfinite(v) and 1e-3 < v < 0.1 for v in lat)
    assert db[search_space.get_model_hash(ARCH)][0] < db[search_space.get_model_hash([[0, 0], [0, 0, 0], [0, 0, 0, 0]])][0]
