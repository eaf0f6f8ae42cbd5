#!/usr/bin/env python3
"""Headline benchmark: utterances/s and p50 forward latency of the HIP forward pass.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]/[2], the configuration the metric is quoted on): arch_vec
[[1,0],[1,0,0],[1,0,0,0]], use_rnn=True, fp32, synthetic filterbank batch x ~ N(0,1) of shape (64, 80, 1000) PER GPU,
random-init weights from the keyed generator ('lively': He-uniform so activations are O(1) at every depth -- the
reference's Xavier init makes this no-skip architecture's activations decay to 1e-25).
A step = one forward over one batch, inputs resident in HBM; with N > 1 every rank forwards its own 64 utterances
(batch-sharded, weak scaling) and the step ends with one RCCL all-gather of the logits.  `value` is the throughput of K
back-to-back steps issued through model.forward_async (the latency-bound LSTM tail of step i overlaps the encoder of step
i+1 on a second HIP stream; every step is a complete forward, all K complete inside the timed region);
`value_sequential` / `p50_forward_ms` are the same K steps as plain model(x) calls (no overlap).

`value` is FP32-EMULATED arithmetic on the 16-bit matrix cores for the GEMM-shaped 82 % of the FLOPs (config.dense_scheme /
config.linear_scheme say how: two fp16 terms per fp32 operand, three MFMAs per product, fp32 accumulate -- error vs fp64 at
the level of an exact fp32 evaluation, tests/); `value_strict_f32` / `ms_per_step_strict_f32` are the same K steps in the
same process with every GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: NBASR_DENSE_MODE=f32 NBASR_LINEAR_MODE=f32).
With N > 1 the headline `value` is the STRONG-scaling number (BASELINE's metric is quoted on the global batch B=64, T=1000 at
1/2/4/8 GPUs: the SAME 64 utterances split over the N ranks, `"scaling": "strong"`) and `value_weak` the weak one (64 utterances
per GPU).  With N = 1 `strong_proxy` times the same steps at 64 / 8 = 8 utterances -- one rank's share of an 8-GPU run -- and
`projected_x8` = 8 x its rate / `value` (north_star: >= 6 x).  `--force-collective` runs the RCCL all-gather with one rank too
(`allgather_us`).

One JSON line on stdout (rank 0).  Besides the driver's contract fields it carries
  roofline      the node op of the search space -- since round 3 a whole cell of three grouped Conv1d per launch (grouped_cell_kernel),
                the graded kernel of SURVEY.md 8(d).  Round 4: every launch priced under its OWN roofline -- x0 in, y out, weights
                against 8 TB/s; the three convolutions' flops against the pipe the launch ran on (fp32 vector rate, or the bf16 MFMA
                peak for the matrix-core cell) -- frac = sum of max(byte time, flop time) / HIP-event time of the launches.  The
                figure rounds 1-3 reported (the algorithmic bytes of the three node ops a fused launch replaces / time / 8 TB/s) is
                kept as frac_credited_node_ops; it is not a bound
  roofline_mfma the dense downsample convs: issued 16-bit MFMA flops (3 fp16 or 6 bf16 products per algorithmic fp32 product) vs
                2.5 PFLOP/s (or algorithmic flops vs the 157.3 TFLOP/s fp32 MFMA peak with NBASR_DENSE_MODE=f32)
  cpu_baseline  the CPU oracle (torch-CPU port of the reference's op sequence) on a bounded sample of the workload, host cores:
                1 warm-up + 3 timed forwards of 16 utterances, then one forward of the metric's own batch
  parity        the logits of the TIMED steps against the oracle's (all utterances) and against a float64 evaluation (8 utterances)
                under the rule of tests/cases.py; a violation makes the run exit non-zero after printing the line
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ARCH = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]
ARCHS = {'conv5': ARCH, 'dense-skip': [[3, 1], [4, 1, 1], [2, 1, 1, 1]]}      # BASELINE configs[1..3] / configs[3]
BATCH, FRAMES, FEATURES = 64, 1000, 80
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3    # ibid.: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0   # ibid.: bf16 MFMA dense peak


def grouped_conv_bytes(batch, channels, frames, kernel, n_skips, groups=100, elem=4):
    """Algorithmic HBM bytes of one fused grouped-conv launch: read x, write y, read each
    fused skip input (`elem` bytes per activation: 4 fp32, 2 bf16), read the fp32 weights + bias once."""
    return float(elem) * batch * channels * frames * (2 + n_skips) + 4.0 * (channels * (channels // groups) * kernel + channels)


def grouped_conv_flops(batch, channels, frames, kernel, groups=100):
    """Algorithmic flops of one grouped-conv launch: 2 (C / groups) k C T B."""
    return 2.0 * (channels // groups) * kernel * channels * frames * batch


def dense_conv_flops(batch, c_in, c_out, kernel, frames_out):
    return 2.0 * batch * frames_out * c_out * c_in * kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=FRAMES)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: --batch utterances per GPU (default); strong: --batch utterances in total, split over the ranks')
    ap.add_argument('--arch', choices=sorted(ARCHS), default='conv5',
                    help="conv5: the benchmark architecture (default); dense-skip: BASELINE configs[3]'s architecture, run in fp32")
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32',
                    help='f32 (default); bf16: BASELINE configs[3] -- activations and GEMM operands stored as bfloat16 '
                         '(model.to(torch.bfloat16)), fp32 accumulation')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='time plain back-to-back model(x) calls only')
    ap.add_argument('--no-strict', action='store_true', help='skip the exact-fp32-MFMA leg (value_strict_f32)')
    ap.add_argument('--no-strong', action='store_true', help='with N > 1: skip the strong-scaling leg (the headline `value` then stays weak)')
    ap.add_argument('--no-strong-proxy', action='store_true',
                    help='with N = 1: skip the strong-scaling proxy (the same steps at --batch / 8 utterances: one rank\'s share at 8 GPUs)')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='chains of pipelined steps in flight at once: W streams (one host thread each) take the K steps in turn, so '
                         'that one step\'s kernels fill the compute units another step\'s leave idle (8 utterances occupy 200 of 256 '
                         'CUs with one wave per SIMD).  Every step is still a complete forward of the whole batch; 1 = one chain')
    ap.add_argument('--tail-group', default='1',
                    help="consecutive steps of a chain share ONE LSTM recurrence + head over all their utterances (model.forward_many): "
                         "1 = off (default: every timed step is a forward of its own from end to end), 'auto' = groups of up to 64 "
                         "utterances (8 steps at 8 per GPU, none at 64), an integer = that many steps.  Bit-identical logits; measured "
                         "+3.5 % at 8 utterances per GPU, +2.6 % at 16 / 32, +0.8 % at 64")
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default='nccl',
                    help='process-group backend with N > 1 (nccl = RCCL, the measured configuration).  gloo: a functional check of the N-rank '
                         'control flow on a box with fewer GPUs than ranks -- ranks then share devices (local rank modulo the device count)')
    ap.add_argument('--force-collective', action='store_true',
                    help='take the RCCL path with one rank too: a 1-rank nccl group on this GPU, every step ends with the all-gather of '
                         'the logits (parallel.ShardedForward(force_collective=True)); reports allgather_us')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs a torch.distributed.run launch with --nproc-per-node {args.gpus}')
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a HIP device (no CPU path in nb_asr_amd)')

    import nb_asr_amd as nb
    from nb_asr_amd.weights import keyed_fill_, keyed_input
    from nb_asr_amd.parallel import ShardedForward

    if args.scaling == 'strong':
        from nb_asr_amd.parallel import shard_bounds
        lo, hi = shard_bounds(args.batch, world, rank)
        global_batch, args.batch = args.batch, hi - lo
        if args.batch * world != global_batch:
            sys.exit(f'--scaling strong needs --batch divisible by the number of GPUs ({global_batch} over {world})')
    device = torch.device('cuda', local_rank if args.backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    # RCCL process group when world > 1 (or --force-collective: a 1-rank nccl group, so that the collective path runs on one GPU too)
    runner = ShardedForward(world_size=world, rank=rank, device=device, force_collective=args.force_collective,
                            backend=args.backend if (world > 1 or args.force_collective) else None)

    arch = ARCHS[args.arch]
    model = nb.get_model(arch, use_rnn=True, dropout_rate=0.0)
    keyed_fill_(model, seed=1235, mode='lively')
    model = model.to(device).eval()
    x = keyed_input(args.batch, args.frames, seed=rank).to(device)           # resident in HBM before timing
    if args.dtype == 'bf16':
        model, x = model.to(torch.bfloat16), x.to(torch.bfloat16)
        args.no_strict = True

    cur = {'x': x}

    def step():
        with torch.no_grad():
            return runner.forward(model, cur['x'])                            # forward + all-gather of logits (N > 1)

    def timed(run_steps):
        runner.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_steps()
        torch.cuda.synchronize()
        runner.barrier()
        return runner.max_over_ranks(time.perf_counter() - t0), out

    def sequential():
        out = None
        for _ in range(args.steps):
            out = step()
        return out

    tail_group = args.tail_group if args.tail_group == 'auto' else int(args.tail_group)

    def pipelined(ways=None):
        # back-to-back batches: the LSTM + head of step i run on a side stream while the main stream already runs the
        # encoder of step i+1 (model.forward_async); every step is a complete forward and all K finish before the
        # closing synchronize
        ways = args.in_flight if ways is None else ways
        if ways <= 1:
            with torch.no_grad():
                handles = [model.forward_async(cur['x']) for _ in range(args.steps)]
                out = None
                for h in handles:
                    out = runner.gather_logits(h.result())
            return out
        # W chains in flight (round 6, model.forward_many): thread i enqueues steps i, i + W, ... on its own stream (its own plan, tapes
        # and cached recurrence graphs: executor.PlanPool keeps a plan with the stream it was released on) and waits for that stream.
        # The all-gathers are issued afterwards by THIS thread in one fixed order (a collective must be enqueued in the same order
        # on every rank).
        if os.environ.get('NBASR_BENCH_FAIL_CHAINS') == '1':          # test hook: the one-GPU fall-back to one chain (in_flight_error)
            raise RuntimeError('NBASR_BENCH_FAIL_CHAINS=1')
        out = None
        for o in model.forward_many([cur['x']] * args.steps, in_flight=ways, tail_group=tail_group):
            out = runner.gather_logits(o)
        return out

    for _ in range(args.warmup):
        out = step()
    with torch.no_grad():
        for _ in range(2):
            model.forward_async(x).result()
    in_flight_error = []

    def warm_chains():
        if args.in_flight > 1 and not args.no_pipeline:
            if world == 1 and not runner.collective and not in_flight_error:
                # one GPU, no collective in the step: if the chains cannot run on this box (a stream that cannot be created, a capture the
                # runtime refuses) the line says so and carries the one-chain number, instead of no line at all.  With N > 1 a rank must
                # not change its sequence of collectives on its own: there the failure stays a failure.
                try:
                    pipelined()
                except Exception as e:          # noqa: BLE001
                    in_flight_error.append(f'{type(e).__name__}: {e}'[:400])
                    args.in_flight = 1
                    model._plans.clear()
                    return
            for _ in range(3):                            # every chain's plan, launch tapes (recorded at a key's second sight) and recurrence
                pipelined()                               # graphs (captured at the third) exist before a timed region: with tail groups a
                                                          # (member, slot) key comes up only once or twice in a pass of K steps

    warm_chains()
    elapsed_seq, out = timed(sequential)
    assert out.shape == (args.batch * world, (((args.frames + 1) // 2) + 1) // 2, 49) and bool(torch.isfinite(out.float()).all())
    if args.no_pipeline:
        elapsed = elapsed_one = elapsed_seq
    else:
        elapsed, out2 = timed(pipelined)
        assert torch.equal(out2, out)
        elapsed_one = elapsed
        if args.in_flight > 1:
            elapsed_one, out3 = timed(lambda: pipelined(1))          # the same K steps as ONE chain, beside the headline
            assert torch.equal(out3, out)

    # p50 forward latency: each forward bracketed by HIP events on the launch stream
    lat = []
    for _ in range(max(args.steps, 5)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        e1.synchronize()
        lat.append(e0.elapsed_time(e1))
    p50 = statistics.median(lat)

    plan = model._plans.values()[-1]
    dense_schemes = dict(plan.dense_schemes)

    # the same K steps with every GEMM on the exact-fp32 MFMA (no operand splitting): the strict-fp32 number
    strict = None
    if not args.no_strict and plan.dense_mode != 'f32':
        saved = {k: os.environ.get(k) for k in ('NBASR_DENSE_MODE', 'NBASR_LINEAR_MODE')}
        os.environ['NBASR_DENSE_MODE'] = os.environ['NBASR_LINEAR_MODE'] = 'f32'
        try:
            model._plans.clear()                          # the modes are read when a plan is built
            for _ in range(2):
                step()
            with torch.no_grad():
                model.forward_async(x).result()
            warm_chains()
            elapsed_strict, out_s = timed(sequential if args.no_pipeline else pipelined)
            worst = float(((out_s.double() - out.double()).abs() / (1e-5 + 1e-4 * out.double().abs())).max())
            strict = {'value': args.batch * world * args.steps / elapsed_strict, 'ms_per_step': 1e3 * elapsed_strict / args.steps,
                      'worst_err_over_tol_vs_default_path': worst, 'logits': out_s.detach().clone()}
            if rank == 0 and not args.no_roofline:
                # the dense convs of THIS leg against the fp32 MFMA peak (v_mfma_f32_32x32x2_f32, no operand splitting)
                strict['roofline_mfma'] = roofline_leg(model, x, args).get('roofline_mfma')
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            model._plans.clear()
            for _ in range(2):
                step()                                    # rebuild the default plan for the legs below
            with torch.no_grad():
                model.forward_async(x).result()
        plan = model._plans.values()[-1]

    # strong scaling (N > 1): the SAME global batch of --batch utterances split over the ranks
    strong = None
    if world > 1 and args.scaling == 'weak' and not args.no_strong and args.batch % world == 0:
        from nb_asr_amd.parallel import shard_bounds
        lo, hi = shard_bounds(args.batch, world, rank)
        cur['x'] = keyed_input(args.batch, args.frames, seed=0).to(device)[lo:hi].contiguous().to(x.dtype)
        for _ in range(3):
            step()
        with torch.no_grad():
            model.forward_async(cur['x']).result()
        warm_chains()
        elapsed_strong, out_g = timed(sequential if args.no_pipeline else pipelined)
        assert out_g.shape[0] == args.batch
        elapsed_strong1 = elapsed_strong
        if args.in_flight > 1 and not args.no_pipeline:
            elapsed_strong1, out_g1 = timed(lambda: pipelined(1))
            assert torch.equal(out_g1, out_g)
        strong = {'value': args.batch * args.steps / elapsed_strong, 'ms_per_step': 1e3 * elapsed_strong / args.steps,
                  'value_one_in_flight': args.batch * args.steps / elapsed_strong1, 'ms_per_step_one_in_flight': 1e3 * elapsed_strong1 / args.steps,
                  'per_gpu_batch': hi - lo, 'global_batch': args.batch}
        cur['x'] = x

    # strong-scaling PROXY on one GPU: the same steps at --batch / 8 utterances = what ONE rank of an 8-GPU
    # strong-scaled run of this batch computes per step (same model, same data: the first eighth of the batch).  8 x its rate over
    # this GPU's rate at the full batch is the speed-up 8 GPUs would reach if the all-gather were free (north star: >= 6 x)
    proxy = None
    if world == 1 and args.scaling == 'weak' and not args.no_strong_proxy and args.batch >= 8 and args.batch % 8 == 0:
        b8 = args.batch // 8
        cur['x'] = x[:b8].contiguous()
        for _ in range(3):
            step()
        with torch.no_grad():
            for _ in range(3):
                model.forward_async(cur['x']).result()
        warm_chains()
        elapsed_p, out_p = timed(sequential if args.no_pipeline else pipelined)
        assert out_p.shape[0] == b8 and bool(torch.isfinite(out_p.float()).all())
        elapsed_p1 = elapsed_p
        if args.in_flight > 1 and not args.no_pipeline:
            elapsed_p1, out_p1 = timed(lambda: pipelined(1))
            assert torch.equal(out_p1, out_p)
        proxy = {'per_gpu_batch': b8, 'value': b8 * args.steps / elapsed_p, 'ms_per_step': 1e3 * elapsed_p / args.steps,
                 'value_one_in_flight': b8 * args.steps / elapsed_p1, 'ms_per_step_one_in_flight': 1e3 * elapsed_p1 / args.steps,
                 'projected_x8': 8.0 * (b8 * args.steps / elapsed_p) / (args.batch * args.steps / elapsed),
                 'note': f'one GPU, the first {b8} of the {args.batch} utterances per step: the per-rank share of an 8-GPU strong-scaled run; '
                         'projected_x8 = 8 x this rate / `value` (all-gather not included: allgather_us with --force-collective)'}
        cur['x'] = x
        for _ in range(2):
            step()                                        # back to the full batch for the legs below
        with torch.no_grad():
            model.forward_async(x).result()

    # the one collective of the path, on its own: HIP events around the all-gather of this rank's logits shard (RCCL when the group is nccl)
    allgather_us = None
    if runner.collective:
        shard = out[: args.batch].contiguous() if out.shape[0] != args.batch else out
        for _ in range(5):
            runner.gather_logits(shard)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            runner.gather_logits(shard)
        e1.record()
        e1.synchronize()
        allgather_us = runner.max_over_ranks(1e3 * e0.elapsed_time(e1) / 20)

    from nb_asr_amd import hip as nb_hip
    weak_value, weak_ms = args.batch * world * args.steps / elapsed, 1e3 * elapsed / args.steps
    # BASELINE's metric is quoted on the GLOBAL batch (B=64, T=1000) at 1/2/4/8 GPUs: with N > 1 the headline `value` is the
    # strong-scaling rate (the same --batch utterances split over the ranks) and the weak one (--batch per GPU) sits beside it
    headline_strong = strong is not None
    result = {
        'metric': 'utterances_per_sec',
        'value': strong['value'] if headline_strong else weak_value,
        'unit': 'utterances/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': strong['ms_per_step'] if headline_strong else weak_ms,
        'value_weak': weak_value if world > 1 and args.scaling == 'weak' else None,
        'ms_per_step_weak': weak_ms if world > 1 and args.scaling == 'weak' else None,
        'in_flight': max(args.in_flight, 1) if not args.no_pipeline else 1,
        'in_flight_error': in_flight_error[0] if in_flight_error else None,
        'tail_group': (args.tail_group if args.in_flight > 1 and not args.no_pipeline else 1),
        'value_one_in_flight': args.batch * world * args.steps / elapsed_one,
        'ms_per_step_one_in_flight': 1e3 * elapsed_one / args.steps,
        'strong_proxy': proxy,
        'allgather_us': allgather_us,
        'p50_forward_ms': p50,
        'value_b_over_p50': args.batch * world / (1e-3 * p50),          # SURVEY 8(d)'s definition: B / p50 of a single forward
        'value_sequential': args.batch * world * args.steps / elapsed_seq,
        'ms_per_step_sequential': 1e3 * elapsed_seq / args.steps,
        'higher_is_better': True,
        'scaling': 'strong' if headline_strong else args.scaling,
        'vs_baseline': None,
        'dtype': args.dtype,
        'value_strict_f32': strict['value'] if strict else None,
        'ms_per_step_strict_f32': strict['ms_per_step'] if strict else None,
        'strict_f32_vs_default_worst_err_over_tol': strict['worst_err_over_tol_vs_default_path'] if strict else None,
        'roofline_mfma_strict_f32': strict.get('roofline_mfma') if strict else None,
        'value_strong': strong['value'] if strong else (args.batch * world * args.steps / elapsed if world == 1 or args.scaling == 'strong' else None),
        'ms_per_step_strong': strong['ms_per_step'] if strong else (1e3 * elapsed / args.steps if world == 1 or args.scaling == 'strong' else None),
        'strong': strong,
        'build_id': nb_hip.build_id(),
        'data': 'synthetic N(0,1) filterbanks (B,80,T) from a keyed generator; random-init He-uniform weights (keyed)',
        'config': {'workload': ('BASELINE configs[1]/[2]: arch_vec [[1,0],[1,0,0],[1,0,0,0]] use_rnn=True, HIP conv + HIP LSTM'
                                if args.arch == 'conv5' else f'arch_vec {arch} (BASELINE configs[3] architecture) use_rnn=True')
                               + (' -- bf16 storage (activations, GEMM operands), fp32 accumulation' if args.dtype == 'bf16' else ' -- fp32'),
                   'per_gpu_batch': strong['per_gpu_batch'] if headline_strong else args.batch,
                   'global_batch': strong['global_batch'] if headline_strong else args.batch * world,
                   'weak_leg': {'per_gpu_batch': args.batch, 'global_batch': args.batch * world} if headline_strong else None,
                   'frames': args.frames, 'features': FEATURES,
                   'parallelism': (f'batch-sharded x{world}, one RCCL all-gather of logits' if world > 1 else
                                   'single GPU, 1-rank RCCL group: every step ends with the all-gather' if runner.collective else 'single GPU'),
                   'pipelined': not args.no_pipeline,
                   'arithmetic': ('fp32 storage and accumulation; `value`: GEMM operands split into 16-bit terms on the 16-bit matrix '
                                  'cores (fp32-emulated, see dense_scheme / linear_scheme); `value_strict_f32`: exact-fp32 MFMA')
                                 if args.dtype == 'f32' else
                                 ('bf16 storage of activations and dense-conv operands (one bf16 MFMA per product), fp32 accumulation, '
                                  'LayerNorm statistics, LSTM and head in fp32; logits returned as bf16'),
                   'dense_scheme': {f'conv_{k}': v for k, v in sorted(dense_schemes.items())},
                   'dense_scheme_legend': 'f16x2 = 2 fp16 terms per operand, 3 v_mfma_f32_16x16x32_f16 per fp32 product; '
                                          'f16x2-image = the same with the LayerNorm writing the pre-split operand; '
                                          'bf16x3 = 3 bf16 terms, 6 MFMAs; f32 = v_mfma_f32_32x32x2_f32',
                   'linear_scheme': plan.linear_mode, 'cell_fusion': bool(plan.cell_fusion),
                   'nbasr_env': {k: v for k, v in sorted(os.environ.items()) if k.startswith('NBASR_')}},
    }

    if rank == 0 and not args.no_roofline:
        result.update(roofline_leg(model, x, args))
    parity_failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # `out` = the logits of the timed steps (the pipelined leg's are asserted bit-equal to them above)
        result['cpu_baseline'], result['parity'], result['parity_strict_f32'] = cpu_baseline_leg(
            model, args, x.detach().float().cpu(), out, strict['logits'] if strict else None)
        parity_failed = not result['parity']['ok'] or not (result['parity_strict_f32'] or {'ok': True})['ok']
    runner.barrier()
    runner.close()
    # RCCL writes its version banner to the C-level stdout buffer, which would otherwise be flushed at exit, AFTER the result:
    # drain it first so that the JSON line is the last thing this process prints
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(result), flush=True)
        if parity_failed:
            sys.exit('bench.py: the timed workload violates the parity rule (see "parity" in the line above)')


def kernel_source_hash():
    """Hash of the graded kernel's sources (grouped_conv.hip, grouped_conv_osplit.hip, grouped_conv_ring.hip, grouped_cell.hip, common.h) and of the table
    that picks the variant per launch: ties a PMC summary to a build."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for name in ('csrc/grouped_conv.hip', 'csrc/grouped_conv_osplit.hip', 'csrc/grouped_conv_ring.hip', 'csrc/grouped_cell.hip', 'csrc/grouped_cell_mfma.hip',
                 'csrc/common.h', 'gc_variant_table.json'):
        with open(os.path.join(here, 'nb_asr_amd', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(kernel_prefix, suffix=''):
    """HBM bytes per launch of a kernel family from the newest committed PMC summary (profiles/rNN_pmc_hbm_traffic.csv:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied by tools/summarize_pmc.py).
    Launch-weighted mean over the family's template instances; None when no summary is committed."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, 'profiles', f'r*_pmc_hbm_traffic{suffix}.csv')))      # ('' = the default workload; '_cfg3_bf16')
    if not files:
        return None, None
    # the counters describe ONE build of the kernel: the summary's first line records the hash of the kernel's sources
    # (tools/summarize_pmc.py); a summary taken on other sources is stale and must not sit next to a fresh `achieved`
    with open(files[-1]) as f:
        first = f.readline().strip()
    if not first.startswith('# grouped_conv_src=') or first.split('=', 1)[1] != kernel_source_hash():
        return None, f'{os.path.basename(files[-1])} is stale for this build (kernel sources changed since the PMC capture)'
    tot = n = 0.0
    with open(files[-1], newline='') as f:
        for row in csv.DictReader(line for line in f if not line.startswith('#')):
            if row['kernel'].startswith(kernel_prefix):
                tot += float(row['hbm_total_MB']) * 1e6 * int(row['launches'])
                n += int(row['launches'])
    return (tot / n if n else None), os.path.basename(files[-1])


def roofline_leg(model, x, args):
    """Re-run `steps` forwards with HIP events around every launch of the graded kernels (same stream)."""
    plan = model._plans.values()[-1]
    plan.timer = []
    with torch.no_grad():
        for _ in range(args.steps):
            model(x)
    torch.cuda.synchronize()
    records, plan.timer = plan.timer, None
    agg = {}
    for kind, meta, e0, e1 in records:
        a = agg.setdefault((kind, meta), [0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
    out = {}
    # the node op of the search space (grouped Conv1d + bias + ReLU + clamp + skip sum): per node, or a whole cell per launch
    elem_b = 2 if args.dtype == 'bf16' else 4
    tot_ms = tot_own_s = tot_own_bytes = tot_own_flop_s = tot_credit_bytes = tot_flops = 0.0
    per_block, launches, kinds = {}, 0, set()
    for (kind, meta), (ms, n) in agg.items():
        if kind == 'grouped_conv':
            blk, c, _, k, frames, n_skips = meta
            ks, skips = (k,), (n_skips,)
        elif kind in ('grouped_cell', 'grouped_cell_mfma'):
            blk, c, ks, skips, frames, _ = meta
        else:
            continue
        kinds.add(kind)
        # what the launch itself has to move and to compute (VERDICT r3 next 4b / ADVICE r3: the launch's OWN roofline is the headline):
        # a node launch: x in, y out, its skip inputs, weights; a fused cell: x0 in, y out, the weights of its three nodes -- x1 / x2
        # never leave the compute unit, so the bytes of the three node operations it replaces are not a bound for it
        credited = sum(grouped_conv_bytes(args.batch, c, frames, kj, sj, elem=elem_b) for kj, sj in zip(ks, skips))
        own = credited if kind == 'grouped_conv' else 2.0 * elem_b * args.batch * c * frames + 4.0 * sum(c * (c // 100) * kj + c for kj in ks)
        fl = sum(grouped_conv_flops(args.batch, c, frames, kj) for kj in ks)
        # the pipe the launch ACTUALLY ran on (from the recorded kernel, not from the plan's switch): the bf16 matrix-core cell is
        # priced against the dense bf16 MFMA peak, everything else against the fp32 vector rate (= the fp32 matrix rate, 157.3 TF)
        flop_peak = BF16_MFMA_PEAK_TFLOPS if kind == 'grouped_cell_mfma' else FP32_MFMA_PEAK_TFLOPS
        t_hbm, t_alu = own / (HBM_PEAK_GBS * 1e9), fl / (flop_peak * 1e12)
        tot_ms += ms
        launches += n
        tot_own_s += max(t_hbm, t_alu) * n
        tot_own_flop_s += (max(t_hbm, t_alu) * n) if t_alu > t_hbm else 0.0
        tot_own_bytes += own * n
        tot_credit_bytes += credited * n
        tot_flops += fl * n
        k_tag = 'x'.join(str(v) for v in ks)
        skip_tag = '-'.join(str(v) for v in skips)
        e = per_block.setdefault(f'block{blk}_C{c}_T{frames}_k{k_tag}_s{skip_tag}_{kind}',
                                 {'own_bytes': own, 'credited_bytes': credited, 'flops': fl, 'ms': 0.0, 'n': 0, 'flop_peak': flop_peak})
        e['ms'] += ms
        e['n'] += n
    if launches:
        prefix = ('nbasr::grouped_cell_mfma_kernel' if 'grouped_cell_mfma' in kinds else
                  'nbasr::grouped_cell_kernel' if 'grouped_cell' in kinds else 'nbasr::grouped_conv_f32')
        # the committed counters describe two workloads in the default modes: the benchmark's own and BASELINE configs[3] per GPU in bf16
        default_wl = (args.batch, args.frames, args.arch, args.dtype) == (BATCH, FRAMES, 'conv5', 'f32')
        cfg3_wl = (args.batch, args.frames, args.arch, args.dtype) == (32, 1600, 'dense-skip', 'bf16')
        traffic, traffic_src = pmc_traffic_per_launch(prefix, '' if default_wl else '_cfg3_bf16')
        if not (default_wl or cfg3_wl) or any(k.startswith('NBASR_') for k in os.environ):
            traffic, traffic_src = None, None
        secs = tot_ms * 1e-3

        def block_entry(v):
            t_us = 1e3 * v['ms'] / v['n']
            t_hbm_us, t_alu_us = 1e6 * v['own_bytes'] / (HBM_PEAK_GBS * 1e9), 1e6 * v['flops'] / (v['flop_peak'] * 1e12)
            return {'us_per_launch': t_us, 'own_bytes_per_launch': v['own_bytes'], 'flops_per_launch': v['flops'],
                    'GBps': v['own_bytes'] / t_us / 1e3, 'TFLOPs': v['flops'] / t_us / 1e6,
                    'bound': 'hbm' if t_hbm_us >= t_alu_us else ('bf16-mfma' if v['flop_peak'] != FP32_MFMA_PEAK_TFLOPS else 'fp32-alu'),
                    'attainable_us': max(t_hbm_us, t_alu_us), 'frac': max(t_hbm_us, t_alu_us) / t_us,
                    'credited_node_ops_GBps': v['credited_bytes'] / t_us / 1e3}
        compute_bound = tot_own_flop_s > 0.5 * tot_own_s
        flop_peak_head = BF16_MFMA_PEAK_TFLOPS if 'grouped_cell_mfma' in kinds else FP32_MFMA_PEAK_TFLOPS
        # 'mfma' only where the launches ran on the matrix pipe (bf16 cell); the fp32 cell is v_pk_fma_f32 work: the fp32 VECTOR ALU
        # (same 157.3 TFLOP/s as the fp32 matrix rate, another pipe)
        head = ({'bound': 'mfma' if 'grouped_cell_mfma' in kinds else 'fp32-alu', 'achieved': tot_flops / secs / 1e12, 'peak': flop_peak_head, 'unit': 'TFLOP/s',
                 'bound_note': 'most of these launches are flop-bound under their own roofline -- x0 in, y out, weights against 8 TB/s; the '
                               "three convolutions' flops against " + ('the dense bf16 MFMA peak (matrix-core cell)' if 'grouped_cell_mfma' in kinds
                                                                       else 'the fp32 vector rate (= the fp32 matrix rate, 157.3 TF)')}
                if compute_bound else {'bound': 'hbm', 'achieved': tot_own_bytes / secs / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'})
        out['roofline'] = {
            'kernel': 'grouped_cell_kernel<T,CG,KEEP1,NTB,GPW> where a cell is three grouped convs and a row fits a workgroup (one launch = the '
                      'three node ops of a cell, x1 / x2 never leave the CU, LayerNorm statistics as a by-product); otherwise '
                      'grouped_conv_f32_{ring_,pipe_,osplit_,}kernel<CG,K,D,..> (variant per launch from gc_variant_table.json; bf16: '
                      'grouped_conv_kernel<bf16_t,..>): fused pad + grouped Conv1d + bias + ReLU + clamp + skip sum [+ LayerNorm on load]; '
                      'bf16 storage: grouped_cell_mfma_kernel<CP,GPW> (the same cell on v_mfma_f32_16x16x32_bf16)',
            'kernels_timed': sorted(kinds),
            **head,
            # frac = sum over launches of max(own bytes / 8 TB/s, flops / peak of the pipe it ran on) / measured time (HIP events)
            'frac': tot_own_s / secs,
            'frac_note': 'each launch under its OWN roofline (round 4; rounds 1-3 credited a fused cell with the algorithmic bytes of the three '
                         'node operations it replaces: frac_credited_node_ops, which is not a bound and can exceed 1 per block)',
            'frac_credited_node_ops': tot_credit_bytes / secs / 1e9 / HBM_PEAK_GBS,
            'credited_node_ops_GBps': tot_credit_bytes / secs / 1e9,
            'hbm_GBps': tot_own_bytes / secs / 1e9, 'frac_of_hbm_peak': tot_own_bytes / secs / 1e9 / HBM_PEAK_GBS,
            'TFLOPs': tot_flops / secs / 1e12, 'frac_of_flop_peak': tot_flops / secs / 1e12 / flop_peak_head,
            'traffic': traffic, 'traffic_source': traffic_src,
            'traffic_source_build': (f'committed PMC capture of the graded kernels at source hash {kernel_source_hash()} (= this build: a capture '
                                     'on other sources is rejected)') if traffic else None,
            'traffic_GBps': (traffic / (1e3 * tot_ms / launches) / 1e3) if traffic else None,
            'own_bytes_per_launch_avg': tot_own_bytes / launches, 'attainable_us_per_launch_avg': 1e6 * tot_own_s / launches,
            'us_per_launch_avg': 1e3 * tot_ms / launches, 'launches_per_forward': launches // args.steps,
            'per_block': {k: block_entry(v) for k, v in sorted(per_block.items())},
        }
    # dense downsample convs (MFMA-bound)
    tot_flops = tot_ms = 0.0
    per_layer = {}
    for (kind, meta), (ms, n) in agg.items():
        if kind != 'dense_conv':
            continue
        blk, c_in, c_out, k, frames_out, _ = meta
        f = dense_conv_flops(args.batch, c_in, c_out, k, frames_out)
        tot_flops += f * n
        tot_ms += ms
        per_layer[f'conv_{blk}_{c_in}x{c_out}_T{frames_out}'] = {'algorithmic_TFLOPs': f * n / (ms * 1e-3) / 1e12, 'us_per_launch': 1e3 * ms / n}
    if tot_ms:
        algorithmic = tot_flops / (tot_ms * 1e-3) / 1e12
        if plan.dense_mode != 'f32':
            # fp32-accurate operand splitting: 6 bf16 MFMA products (3-way bf16 split) or 3 fp16 MFMA products (2-way fp16
            # split) are issued per algorithmic fp32 product; bf16 and fp16 MFMAs have the same dense peak
            issued_flops = 0.0
            for (kind, meta), (ms, n) in agg.items():
                if kind == 'dense_conv':
                    scheme = plan.dense_schemes.get(meta[0], 'bf16x3')
                    per_layer[f'conv_{meta[0]}_{meta[1]}x{meta[2]}_T{meta[4]}']['scheme'] = scheme
                    if meta[0] in plan.dense_row_tiles:
                        per_layer[f'conv_{meta[0]}_{meta[1]}x{meta[2]}_T{meta[4]}']['row_tile'] = plan.dense_row_tiles[meta[0]]
                        per_layer[f'conv_{meta[0]}_{meta[1]}x{meta[2]}_T{meta[4]}']['frame_tile'] = plan.dense_frame_tiles.get(meta[0], 256)
                    issued_flops += (1.0 if scheme == 'bf16' else 3.0 if scheme.startswith('f16x2') else 6.0) * dense_conv_flops(args.batch, meta[1], meta[2], meta[3], meta[4]) * n
            issued = issued_flops / (tot_ms * 1e-3) / 1e12
            out['roofline_mfma'] = {'kernel': 'gemm_conv_split_kernel<P,S> (dense k=8 conv on v_mfma_f32_16x16x32_{f16,bf16}: 3 fp16 or 6 bf16 products per fp32 product)',
                                    'bound': 'mfma', 'achieved': issued, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s (16-bit MFMA issued)',
                                    'frac': issued / BF16_MFMA_PEAK_TFLOPS, 'traffic': None,
                                    'algorithmic_fp32_TFLOPs': algorithmic, 'vs_fp32_mfma_peak': algorithmic / FP32_MFMA_PEAK_TFLOPS,
                                    'per_layer': per_layer}
        else:
            out['roofline_mfma'] = {'kernel': 'gemm_conv_kernel<8,S> (dense k=8 conv, v_mfma_f32_32x32x2_f32)', 'bound': 'mfma',
                                    'achieved': algorithmic, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': algorithmic / FP32_MFMA_PEAK_TFLOPS, 'traffic': None, 'per_layer': per_layer}
    # where the forward's time goes (event-bracketed launches, ms per forward)
    split = {}
    for (kind, _meta), (ms, _n) in agg.items():
        split[kind] = split.get(kind, 0.0) + ms / args.steps
    out['ms_per_forward_by_kernel'] = split
    return out


def cpu_baseline_leg(model, args, x_cpu, got, got_strict=None):
    """The CPU oracle (torch-CPU port of the reference's op sequence) on a bounded sample of the same workload -- and, from the same
    oracle runs, the PARITY of the timed workload: `got` = the logits the timed HIP steps produced for `x_cpu`.

    Timing protocol (>= 1 warm-up + >= 3 timed forwards): one warm-up and three timed forwards at B = min(16, batch);
    `value` is the rate of ONE forward at the metric's own batch when that takes under a minute (it does on the GPU box's host:
    ~20 s), else the small-batch rate.  Parity: the oracle's fp32 logits for every utterance of the batch, its float64 evaluation for
    the first 8, and tests/cases.py's rule (un-relaxed north-star tolerance where the fp32 reference itself is within 0.4 of it against
    fp64; else RMS error against fp64 <= 1.5 x the reference's, worst element <= 2 x; bf16: 1.25 x / 1.5 x the reference's bf16 forward)."""
    from oracle import asr_oracle as oracle
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests'))
    import cases
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    small_b = min(16, args.batch)
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32      # bf16: the reference's model.to(torch.bfloat16) forward
    arch = ARCHS[args.arch]
    fwd = lambda inp: oracle.asr_forward(params, arch, inp, use_rnn=True, dtype=dt)      # noqa: E731
    with torch.no_grad():
        xs = x_cpu[:small_b]
        fwd(xs)                                                                         # warm-up (thread pool, allocator)
        small_times = []
        for _ in range(3):
            t0 = time.perf_counter()
            want_small = fwd(xs)
            small_times.append(time.perf_counter() - t0)
        small = statistics.median(small_times)
        full = small * args.batch / small_b < 60.0
        if full and args.batch != small_b:
            t0 = time.perf_counter()
            want = fwd(x_cpu)
            secs, sample_b = time.perf_counter() - t0, args.batch
        else:
            want, secs, sample_b = want_small, small, small_b
        n64 = min(8, sample_b)
        p64 = {k: v.double() for k, v in params.items()} if dt == torch.float32 else {k: v.float().double() for k, v in params.items()}
        truth = oracle.asr_forward(p64, arch, x_cpu[:n64].float().double(), use_rnn=True, dtype=torch.float64)
    # ---- parity of the timed workload --------------------------------------------------------------------------------------------
    w, t8 = want.float().double(), truth.double()
    w8 = w[:n64]
    noise = cases.worst_ratio(w8, t8, 1e-4, 1e-5)

    def per_utterance_rms(e):
        return e.pow(2).mean(dim=(1, 2)).sqrt()

    def parity_of(got_logits):
        g = got_logits[:sample_b].detach().float().cpu().double()
        ratio_all = cases.worst_ratio(g, w, 1e-4, 1e-5)
        # how far, not only whether: the share of logits outside the UN-relaxed north-star bound against the
        # oracle and the 99.9th percentile of err / tol, over all utterances of the sample
        eot = ((g - w).abs() / (1e-5 + 1e-4 * w.abs())).flatten()
        frac_outside = float((eot > 1.0).double().mean())
        p999 = float(torch.quantile(eot[torch.randperm(eot.numel(), generator=torch.Generator().manual_seed(0))[: 1 << 22]], 0.999)) if eot.numel() else 0.0
        eot_ref = ((w8 - t8).abs() / (1e-5 + 1e-4 * t8.abs())).flatten()      # the oracle itself against fp64, same two figures
        g8 = g[:n64]
        rms_ratio = cases._rms(g8 - t8) / max(cases._rms(w8 - t8), 1e-300)
        max_ratio = float((g8 - t8).abs().max()) / max(float((w8 - t8).abs().max()), 1e-300)
        # every utterance gates: only the first 8 have an fp64 truth, so the others are held to the oracle through the
        # spread the truth-checked ones show -- a tile or batch-index bug at b >= 8 is orders of magnitude, not a factor of two
        dist = per_utterance_rms(g - w)
        spread = float(dist.max()) / max(float(dist[:n64].max()), 1e-300)
        batch_ok = spread <= 2.0
        if args.dtype == 'bf16':
            leg, ok = 'bf16: rms <= 1.25 x, worst <= 1.5 x the reference bf16 forward (vs fp64)', rms_ratio <= 1.25 and max_ratio <= 1.5
        elif noise < cases.QUIET:
            leg, ok = 'quiet: north-star tolerance un-relaxed', cases.worst_ratio(g8, w8, 1e-4, 1e-5) <= 1.0 and ratio_all <= 1.0
        else:
            f_rms, f_max = cases.FACTORS
            leg = f'noisy: rms <= {f_rms} x, worst <= {f_max} x the fp32 reference (vs fp64)'
            ok = rms_ratio <= f_rms and cases.worst_ratio(g8, t8, 1e-4, 1e-5) <= f_max * noise
        return {'ok': bool(ok and batch_ok), 'leg': leg, 'ratio_vs_oracle': ratio_all, 'utterances_vs_oracle': sample_b,
                'frac_outside_bound': frac_outside, 'p999_err_over_tol': p999,
                'oracle_vs_fp64_frac_outside_bound': float((eot_ref > 1.0).double().mean()),
                'oracle_vs_fp64_p999_err_over_tol': float(torch.quantile(eot_ref, 0.999)) if 0 < eot_ref.numel() <= (1 << 24) else None,
                'oracle_noise_vs_fp64': noise, 'rms_ratio': rms_ratio, 'worst_ratio_vs_fp64': max_ratio, 'utterances_vs_fp64': n64,
                'worst_utterance_distance_to_oracle_over_worst_truth_checked': spread,
                'whole_batch_rule': 'per-utterance RMS distance to the oracle, every utterance <= 2 x the largest among the fp64-checked ones',
                'tolerance': 'north star: |err| <= 1e-5 + 1e-4 |ref| (ratio_vs_oracle, oracle_noise_vs_fp64 in units of it); rule: tests/cases.py::assert_parity'}

    parity = parity_of(got)
    # the exact-fp32 leg (every GEMM on v_mfma_f32_*_f32, no operand splitting) under the same rule: how far ANY fp32 evaluation in
    # another summation order sits from the oracle on this workload
    parity_strict = parity_of(got_strict) if got_strict is not None else None
    base = {'value': sample_b / secs, 'unit': 'utterances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'batch': sample_b, 'frames': args.frames, 'forwards_timed': 3 + (1 if sample_b != small_b else 0),
            'sample': f'oracle/asr_oracle.py (torch CPU ops, the reference\'s op sequence): 1 warm-up + 3 timed forwards of B={small_b}, T={args.frames} '
                      f'(median {small:.2f} s: value_small_batch)' + (f', then ONE forward of the metric\'s own batch B={sample_b} ({secs:.1f} s: value)'
                                                                      if sample_b != small_b else ' (value)') + f'; os.cpu_count()={os.cpu_count()}',
            'seconds_per_forward': secs, 'value_small_batch': small_b / small, 'small_batch': small_b,
            'seconds_small_batch': small_times}
    return base, parity, parity_strict


if __name__ == '__main__':
    main()
