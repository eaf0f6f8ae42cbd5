"""BASELINE config 5 on the GPU: `bench_dataset.latency_sweep` at the config's own shape (B=32, T=1000) over architectures
that cover all six main ops, the emitted `nb-asr-bench-{device}.pickle` read back (configs[4] had no -m gpu test).
Format reference: /root/reference/nasbench_asr/dataset.py:28-67, 168-240; file-name rule :484,537,544."""
import math
import pickle

import pytest
import torch

from nb_asr_amd import bench_dataset, search_space
from nb_asr_amd.weights import keyed_input

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# every main op (0 linear, 1 conv5, 2 conv5d2, 3 conv7, 4 conv7d2, 5 zero), skips from none to all
ARCHS = [
    [[1, 0], [1, 0, 0], [1, 0, 0, 0]],        # BASELINE configs 1-3
    [[3, 1], [4, 1, 1], [2, 1, 1, 1]],        # BASELINE config 4 (dense skips)
    [[0, 1], [5, 1, 0], [2, 0, 1, 1]],
    [[0, 0], [0, 1, 0], [0, 0, 0, 1]],        # all linear
    [[5, 1], [3, 0, 1], [1, 1, 0, 0]],
    [[4, 0], [2, 1, 1], [5, 0, 1, 1]],
    [[2, 1], [1, 0, 0], [4, 1, 0, 1]],
    [[3, 0], [0, 1, 1], [3, 1, 1, 0]],
    [[1, 1], [4, 0, 1], [0, 1, 1, 1]],
]


def test_architectures_cover_all_ops_and_are_distinct():
    assert {node[0] for a in ARCHS for node in a} == set(range(6))
    assert len({search_space.get_model_hash(a) for a in ARCHS}) == len(ARCHS) >= 8


def test_latency_sweep_emits_a_loadable_dataset(tmp_path):
    work = [(search_space.get_model_hash(a), a) for a in ARCHS]
    rows = bench_dataset.latency_sweep(work, torch.device(DEV), batch=32, frames=1000, warmup=1, iters=3)
    assert [h for h, _ in rows] == [h for h, _ in work]
    path = tmp_path / bench_dataset.file_name('mi355x-fp32')
    bench_dataset.write_benchmarking_dataset(path, 'mi355x-fp32', sorted(rows), meta={'batch_size': 32, 'frames': 1000})
    with open(path, 'rb') as f:
        header, data = pickle.load(f), pickle.load(f)
        assert f.read() == b''
    assert sorted(header) == ['columns', 'dataset_type', 'device', 'search_space', 'version']
    assert len(data) == len({r[0] for r in data}) == len(ARCHS)                   # one row per unique hash
    hdr, device, db = bench_dataset.read_benchmarking_dataset(path)
    assert device == 'mi355x-fp32'
    lat = {}
    for arch in ARCHS:                                                              # hash <-> architecture, like BenchmarkingDataset.latency
        (latency,) = db[search_space.get_model_hash(arch, ops=hdr['search_space']['ops'])]
        assert math.isfinite(latency) and 1e-3 < latency < 1.0, (arch, latency)     # a 32 x 1000 forward: milliseconds
        lat[str(arch)] = latency
    # sanity of the numbers themselves: an all-`linear` cell (18 x 3 dense C x C GEMMs) costs more than conv5 x 3
    assert lat[str(ARCHS[3])] > 1.3 * lat[str(ARCHS[0])]
    summary = bench_dataset.summarize(rows, dict(work))
    assert summary['architectures'] == len(ARCHS) and set(summary['median_s_by_op_used']) == set(search_space.all_ops)


def test_bank_models_compute_what_ordinary_models_compute():
    """A model whose parameters come from the sweep's WeightBank is an ordinary model: same logits as a model built the
    usual way with the same keyed 'lively' weights (guards the meta-device construction and the parameter plumbing)."""
    import nb_asr_amd as nb
    from nb_asr_amd.weights import keyed_fill_
    arch = [[0, 1], [5, 1, 0], [2, 0, 1, 1]]
    bank = bench_dataset.WeightBank(DEV)
    banked = bank.build(arch)
    plain = keyed_fill_(nb.get_model(arch, use_rnn=True, dropout_rate=0.0), seed=1235, mode='lively').to(DEV).eval()
    x = keyed_input(2, 64, seed=1).to(DEV)
    with torch.no_grad():
        assert torch.equal(banked(x), plain(x))
        other = bank.build([[1, 0], [1, 0, 0], [1, 0, 0, 0]])                       # a second architecture through the same pool
        assert torch.isfinite(other(x)).all()
        assert torch.equal(banked(x), plain(x))
    assert len(bank.pool) == 1
