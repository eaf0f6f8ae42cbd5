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
    bench_dataset.write_benchmarking_dataset(path, 'mi355x-fp32', _rows(), extra_header={'batch_size': 32})
    assert path.name == 'nb-asr-bench-mi355x-fp32.pickle'
    with open(path, 'rb') as f:                              # exactly two pickles: header, then rows
        header, data = pickle.load(f), pickle.load(f)
        assert f.read() == b''
    assert header['dataset_type'] == 'benchmarking' and header['device'] == 'mi355x-fp32' and header['version'] >= 1
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
    bad = tmp_path / 'bad.pickle'
    with open(bad, 'wb') as f:
        pickle.dump({'dataset_type': 'training', 'device': 'x', 'columns': []}, f)
    with pytest.raises(ValueError, match='benchmarking'):
        bench_dataset.read_benchmarking_dataset(bad)


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
    finally:
        sys.path.remove(str(REF))
        for name in [m for m in sys.modules if m.startswith('nasbench_asr')]:
            del sys.modules[name]
