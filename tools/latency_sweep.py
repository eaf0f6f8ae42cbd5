#!/usr/bin/env python3
"""BASELINE config 5: latency of every unique architecture (B=32, T=1000), one architecture per GPU at a time.

    python tools/latency_sweep.py --out DIR [--limit N] [--batch 32] [--frames 1000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/latency_sweep.py --out DIR

Embarrassingly parallel: rank r takes architectures r, r+G, ... of the 8 242 unique ones; no collective on the data path;
rows are gathered once at the end (all_gather_object) and rank 0 writes DIR/nb-asr-bench-{device}.pickle in the format
the reference's BenchmarkingDataset / from_folder load.
"""
import argparse
import os
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import bench_dataset


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--device-name', default='mi355x-fp32')
    ap.add_argument('--limit', type=int, default=None)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--summary', default=None, help='also write the min / median / max + per-op-family summary here (JSON)')
    ap.add_argument('--fill', default='lively', help="parameter values of the sweep's shared WeightBank: 'lively' (keyed He-uniform, default) "
                                                     "or a constant such as 0.01 (round 2's fill, for an A/B)")
    ap.add_argument('--stride', type=int, default=1, help='take every N-th architecture of the work list (A/B samples)')
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)
    import time
    t0 = time.time()
    everything = bench_dataset.sweep_work_list(a.limit)[::a.stride]
    work = everything[rank::world]
    fill = a.fill if a.fill in ('lively', 'xavier') else float(a.fill)
    rows = bench_dataset.latency_sweep(work, device, a.batch, a.frames, warmup=a.warmup, iters=a.iters, progress=500 if rank == 0 else None,
                                       fill=fill)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
        dist.destroy_process_group()
    if rank == 0:
        out = pathlib.Path(a.out)
        out.mkdir(parents=True, exist_ok=True)
        path = out / bench_dataset.file_name(a.device_name)
        from nb_asr_amd import hip
        meta = {'batch_size': a.batch, 'frames': a.frames, 'features': 80, 'dtype': 'fp32', 'n_gpus': world, 'use_rnn': True,
                'protocol': f'{a.warmup} warm-up + median of {a.iters} forwards, HIP events on the launch stream, one architecture '
                            f'per GPU at a time; input x ~ N(0, 1); parameters shared through one WeightBank, fill = {a.fill!r} '
                            f'(lively: weights.keyed_values(key, shape, 1235, "lively") -- He-uniform weights, live biases / gamma / beta)',
                'weight_fill': a.fill, 'build_id': hip.build_id(), 'sweep_seconds': time.time() - t0,
                'gpu': torch.cuda.get_device_name(device)}
        bench_dataset.write_benchmarking_dataset(path, a.device_name, sorted(rows), meta=meta)
        summary = dict(bench_dataset.summarize(rows, dict(everything)), **meta, file=path.name)
        if a.summary:
            import json
            pathlib.Path(a.summary).write_text(json.dumps(summary, indent=1, sort_keys=True) + '\n')
        print(f'wrote {path}: {len(rows)} architectures in {meta["sweep_seconds"]:.0f} s, latency min {summary["latency_min_s"] * 1e3:.2f} ms  '
              f'median {summary["latency_median_s"] * 1e3:.2f} ms  max {summary["latency_max_s"] * 1e3:.2f} ms')


if __name__ == '__main__':
    main()
