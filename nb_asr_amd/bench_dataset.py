"""Latency sweep over the search space and the `nb-asr-bench-{device}.pickle` emitter (BASELINE config 5).

File format = what the reference's ``BenchmarkingDataset`` loads (``nasbench_asr/dataset.py:28-67, 168-240``):
two consecutive pickles in one file -- a header dict, then a list of rows.

    header = {'dataset_type': 'benchmarking', 'device': <name matching [a-zA-Z0-9-]+>, 'version': int,
              'search_space': {'shape': [[6,2],[6,2,2],[6,2,2,2]], 'ops': [...], 'nodes': 3},
              'columns': ['model_hash', 'latency', ...]}
    rows   = [[model_hash, latency_seconds, ...], ...]        # hash from get_model_hash(arch, ops=header ops)

``from_folder`` (``dataset.py:477-555``) picks the file up by its name ``nb-asr-bench-{device}.pickle``.
"""
import pickle
import re
import statistics
import time

from . import search_space

DATASET_VERSION = 1
_DEVICE_RE = re.compile(r'[a-zA-Z0-9-]+')


def make_header(device, columns=('model_hash', 'latency'), ops=None, nodes=None, extra=None):
    if not _DEVICE_RE.fullmatch(device):
        raise ValueError(f'device name {device!r} must match [a-zA-Z0-9-]+ (it becomes part of the file name)')
    columns = list(columns)
    if columns[:2] != ['model_hash', 'latency']:
        raise ValueError("columns must start with ['model_hash', 'latency']")
    ops = list(search_space.all_ops if ops is None else ops)
    nodes = search_space.default_nodes if nodes is None else nodes
    header = {'dataset_type': 'benchmarking', 'device': device, 'version': DATASET_VERSION,
              'search_space': {'shape': search_space.get_search_space(ops, nodes), 'ops': ops, 'nodes': nodes},
              'columns': columns}
    if extra:
        header.update(extra)
    return header


def file_name(device):
    return f'nb-asr-bench-{device}.pickle'


def write_benchmarking_dataset(path, device, rows, columns=('model_hash', 'latency'), extra_header=None):
    """rows: iterable of [model_hash, latency, ...] (one per unique architecture)."""
    header = make_header(device, columns, extra=extra_header)
    rows = [list(r) for r in rows]
    for r in rows:
        if len(r) != len(header['columns']) or not isinstance(r[0], str):
            raise ValueError(f'row {r!r} does not match columns {header["columns"]}')
    with open(path, 'wb') as f:
        pickle.dump(header, f)
        pickle.dump(rows, f)
    return header


def read_benchmarking_dataset(path):
    """Minimal reader with the reference loader's checks: returns (header without 'device', device, {hash: rest})."""
    with open(path, 'rb') as f:
        header = pickle.load(f)
        if header['dataset_type'] != 'benchmarking':
            raise ValueError('Expected a dataset file with benchmarking information')
        device = header.pop('device')
        if header['columns'][:2] != ['model_hash', 'latency']:
            raise ValueError('expected the dataset to contain information in order: model hash, latency')
        data = pickle.load(f)
    return header, device, {h: rest for h, *rest in data}


def sweep_work_list(limit=None, rank=0, world_size=1):
    """(hash, arch) of every unique architecture (first enumerated representative), round-robin over ranks."""
    items = list(search_space.get_unique_architectures().items())
    if limit is not None:
        items = items[:limit]
    return items[rank::world_size]


def build_for_timing(arch_vec, device, use_rnn=True):
    """Model on `device` with cheap constant weights (latency does not depend on weight values; skips the ~1 s
    Xavier initialisation of 26 M parameters that dominates a per-architecture sweep)."""
    import torch
    from .model import ASRModel
    with torch.device(device):
        model = ASRModel(search_space.arch_vec_to_names(arch_vec), use_rnn=use_rnn, dropout_rate=0.0)
    with torch.no_grad():
        for p in model.parameters():
            p.fill_(0.01)
    return model.eval()


def measure_latency(model, x, warmup=2, iters=5):
    """Median wall-clock seconds of one forward (HIP events on the launch stream)."""
    import torch
    with torch.no_grad():
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize(x.device)
        times = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(x)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(times)


def latency_sweep(work, device, batch=32, frames=1000, warmup=2, iters=5, progress=None):
    """[[hash, latency_s], ...] for the (hash, arch) pairs in `work`, one model at a time on `device`."""
    import torch
    x = torch.randn(batch, 80, frames, device=device)
    rows, t0 = [], time.time()
    for i, (h, arch) in enumerate(work):
        model = build_for_timing(arch, device)
        rows.append([h, measure_latency(model, x, warmup, iters)])
        del model
        if progress and (i + 1) % progress == 0:
            print(f'  {i + 1}/{len(work)} architectures, {time.time() - t0:.0f} s', flush=True)
    return rows
