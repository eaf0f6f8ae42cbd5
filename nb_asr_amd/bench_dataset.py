"""Latency sweep over the search space and the `nb-asr-bench-{device}.pickle` emitter (BASELINE config 5).

File format = what the reference's ``BenchmarkingDataset`` loads (``nasbench_asr/dataset.py:28-67, 168-240``):
two consecutive pickles in one file -- a header dict, then a list of rows.

    header = {'dataset_type': 'benchmarking', 'device': <name matching [a-zA-Z0-9-]+>, 'version': int,
              'search_space': {'shape': [[6,2],[6,2,2],[6,2,2,2]], 'ops': [...], 'nodes': 3},
              'columns': ['model_hash', 'latency', ...]}
    rows   = [[model_hash, latency_seconds, ...], ...]        # hash from get_model_hash(arch, ops=header ops)

``from_folder`` (``dataset.py:477-555``) picks the file up by its name ``nb-asr-bench-{device}.pickle``.

The header holds ONLY the keys the reference's files have: its loader pops ``device`` and then requires the remaining
header of every file it is given to be equal (``dataset.py:37-42``), so run metadata (batch, frames, dtype, number of
GPUs, build id, timing protocol) goes to a side-car ``nb-asr-bench-{device}.meta.json`` that ``from_folder``'s file-name
patterns ignore -- a 1-GPU and an 8-GPU sweep, or this file and an upstream device file, load together.
"""
import json
import pickle
import re
import statistics
import time

from . import search_space

DATASET_VERSION = 1
_DEVICE_RE = re.compile(r'[a-zA-Z0-9-]+')


def make_header(device, columns=('model_hash', 'latency'), ops=None, nodes=None):
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
    return header


def file_name(device):
    return f'nb-asr-bench-{device}.pickle'


def meta_name(device):
    return f'nb-asr-bench-{device}.meta.json'


def write_benchmarking_dataset(path, device, rows, columns=('model_hash', 'latency'), meta=None):
    """rows: iterable of [model_hash, latency, ...] (one per unique architecture).  ``meta`` (a dict of run metadata) is
    written next to the pickle as ``nb-asr-bench-{device}.meta.json``, never into the header."""
    header = make_header(device, columns)
    rows = [list(r) for r in rows]
    seen = set()
    for r in rows:
        if len(r) != len(header['columns']) or not isinstance(r[0], str):
            raise ValueError(f'row {r!r} does not match columns {header["columns"]}')
        if r[0] in seen:
            raise ValueError(f'model hash {r[0]} appears twice: one row per unique architecture')
        seen.add(r[0])
    with open(path, 'wb') as f:
        pickle.dump(header, f)
        pickle.dump(rows, f)
    if meta is not None:
        import pathlib
        side = pathlib.Path(path).with_name(meta_name(device))
        side.write_text(json.dumps(dict(meta, device=device, rows=len(rows)), indent=1, sort_keys=True) + '\n')
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


class WeightBank:
    """ONE set of device parameters that every architecture of a sweep borrows from.

    Building 8 242 models one after the other spends its time allocating and initialising 26 M parameters and re-packing
    the same four dense convolutions and the LSTM for the matrix cores, not measuring.  The bank holds, per state_dict key
    and shape, one ``nn.Parameter``; ``build`` constructs the module tree on the ``meta`` device (no allocation, no
    initialisation) and plugs the bank's parameters in.  ``fill='lively'`` (default since round 3): the values
    ``weights.keyed_fill_(model, 1235, 'lively')`` gives that key -- He-uniform weights, non-trivial biases / gamma / beta, so
    activations are O(1) and distinct per channel.  Round 2 filled every weight with the constant 0.01 ("latency does not
    depend on weight values"): all channels were then equal, every LayerNorm output the constant beta and every split's low
    term zero -- and on this chip MFMA clocks DO depend on operand toggling.  ``fill=<float>`` keeps
    that constant fill for an A/B (tools/latency_sweep.py --fill).  All models
    share one ``PlanPool``: workspaces are allocated once, and because the dense / LSTM / head parameters are the SAME
    tensor objects for every architecture, their packed copies are built once for the whole sweep (a `linear` node op's
    weight is shared per (position, shape), too)."""

    def __init__(self, device, use_rnn=True, fill='lively', seed=1235):
        import torch
        from .executor import PlanPool
        self.device, self.use_rnn, self.fill, self.seed = torch.device(device), use_rnn, fill, seed
        self._params = {}
        self.pool = PlanPool()

    def _param(self, key, meta_param):
        import torch
        shape = tuple(meta_param.shape)
        k = (key, shape)
        p = self._params.get(k)
        if p is None:
            if isinstance(self.fill, str):
                from .weights import keyed_values
                values = keyed_values('model.' + key if not key.startswith('model.') else key, shape, self.seed, self.fill).to(self.device)
            else:
                value = 1.0 if (key.endswith('weight') and meta_param.dim() == 1) else float(self.fill)   # LayerNorm gamma = 1
                values = torch.full(shape, value, device=self.device, dtype=torch.float32)
            p = self._params[k] = torch.nn.Parameter(values, requires_grad=False)
        return p

    def build(self, arch_vec):
        import torch
        from .model import ASRModel
        with torch.device('meta'):
            model = ASRModel(search_space.arch_vec_to_names(arch_vec), use_rnn=self.use_rnn, dropout_rate=0.0)
        for mod_name, mod in model.named_modules():
            for name in list(mod._parameters):
                if mod._parameters[name] is not None:
                    mod._parameters[name] = self._param(f'{mod_name}.{name}' if mod_name else name, mod._parameters[name])
        for mod in model.modules():
            if isinstance(mod, torch.nn.LSTM):
                mod._flat_weights = [getattr(mod, n) for n in mod._flat_weights_names]
        model._plans = self.pool
        return model.eval()


def build_for_timing(arch_vec, device, use_rnn=True, bank=None):
    """Model on `device` for latency measurements.  With a ``WeightBank`` the parameters, workspaces and packed weights are
    shared with every other model built from it."""
    return (bank or WeightBank(device, use_rnn)).build(arch_vec)


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


def latency_sweep(work, device, batch=32, frames=1000, warmup=2, iters=5, progress=None, fill='lively'):
    """[[hash, latency_s], ...] for the (hash, arch) pairs in `work`, one model at a time on `device` (BASELINE config 5:
    one architecture per GPU at a time; all models of the sweep borrow their parameters from one WeightBank)."""
    import torch
    x = torch.randn(batch, 80, frames, device=device)
    bank = WeightBank(device, fill=fill)
    rows, t0 = [], time.time()
    for i, (h, arch) in enumerate(work):
        model = bank.build(arch)
        rows.append([h, measure_latency(model, x, warmup, iters)])
        del model
        if progress and (i + 1) % progress == 0:
            print(f'  {i + 1}/{len(work)} architectures, {time.time() - t0:.0f} s', flush=True)
    bank.pool.clear()
    return rows


def summarize(rows, by_hash=None):
    """min / median / max latency (seconds) and, when ``by_hash`` maps hash -> arch_vec, the median per main-op family
    (an architecture counts towards every op it uses in one of its three nodes)."""
    lat = sorted(r[1] for r in rows)
    out = {'architectures': len(rows), 'latency_min_s': lat[0], 'latency_median_s': statistics.median(lat), 'latency_max_s': lat[-1]}
    if by_hash:
        fam = {}
        for h, latency, *_ in rows:
            for op in {search_space.all_ops[node[0]] for node in by_hash[h]}:
                fam.setdefault(op, []).append(latency)
        out['median_s_by_op_used'] = {op: statistics.median(v) for op, v in sorted(fam.items())}
        out['count_by_op_used'] = {op: len(v) for op, v in sorted(fam.items())}
    return out
