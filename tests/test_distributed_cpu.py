"""world_size-2 gloo tests of the batch-sharded runner (nb_asr_amd/parallel.py) on CPU.

The HIP forward itself needs a GPU, so a stand-in per-utterance "model" is used here: what is under test is
the sharding arithmetic, the single all-gather (equal and ragged shards), rank ordering, and the barrier /
max-over-ranks timing helpers bench.py relies on.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from nb_asr_amd.parallel import ShardedForward, shard_bounds


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _fake_model(x):
    """Per-utterance function with the model's output geometry: (b, 80, T) -> (b, ceil(ceil(T/2)/2), 49)."""
    t_out = ((x.shape[2] + 1) // 2 + 1) // 2
    feat = x.mean(dim=1)[:, : t_out * 4: 4]
    return feat.unsqueeze(-1) * torch.arange(1, 50, dtype=x.dtype).view(1, 1, 49) + x.sum(dim=(1, 2)).view(-1, 1, 1)


def _worker(rank, world, port, n_items, out_queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    runner = ShardedForward(world_size=world, rank=rank, device='cpu', backend='gloo')
    torch.manual_seed(0)                                   # same "global batch" on every rank
    global_x = torch.randn(n_items, 80, 37)
    want = _fake_model(global_x)
    got = runner.forward_global(_fake_model, global_x)     # ragged when n_items % world != 0
    ok_global = torch.equal(got, want)
    # bench.py's path: every rank forwards its own equally sized shard, then one all-gather
    local = torch.full((3, 80, 16), float(rank + 1))
    gathered = runner.forward(_fake_model, local)
    ok_local = gathered.shape[0] == 3 * world and all(
        torch.equal(gathered[3 * r: 3 * r + 3], _fake_model(torch.full((3, 80, 16), float(r + 1)))) for r in range(world))
    runner.barrier()
    slowest = runner.max_over_ranks(10.0 + rank)
    runner.close()
    out_queue.put((rank, ok_global, ok_local, slowest))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('n_items', [8, 5])
def test_two_rank_gloo_sharded_forward(n_items):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_global, ok_local, slowest in results:
        assert ok_global, f'rank {rank}: gathered logits differ from the unsharded result'
        assert ok_local, f'rank {rank}: equal-shard all-gather is wrong'
        assert slowest == 10.0 + (world - 1)


def _grad_worker(rank, world, port, out_queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    runner = ShardedForward(world_size=world, rank=rank, device='cpu', backend='gloo')
    torch.manual_seed(0)                                   # the same replica on every rank
    net = torch.nn.Sequential(torch.nn.Linear(7, 300), torch.nn.Tanh(), torch.nn.Linear(300, 5))
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
    unused = torch.nn.Parameter(torch.ones(11))            # no gradient on any rank: must come back as zeros, not hang
    x = torch.randn(6, 7)
    lo, hi = shard_bounds(6, world, rank)
    net(x[lo:hi]).pow(2).sum().backward()                  # this rank's shard; loss = SUM over utterances
    runner.allreduce_gradients(list(net.parameters()) + [frozen, unused], bucket_bytes=4096)     # several buckets
    ref = torch.nn.Sequential(torch.nn.Linear(7, 300), torch.nn.Tanh(), torch.nn.Linear(300, 5))
    ref.load_state_dict(net.state_dict())
    ref(x).pow(2).sum().backward()                         # the whole batch on one rank
    ok = all(torch.allclose(p.grad * world, q.grad, rtol=1e-5, atol=1e-6) for p, q in zip(net.parameters(), ref.parameters()))
    ok = ok and frozen.grad is None and unused.grad is not None and bool((unused.grad == 0).all())
    runner.close()
    out_queue.put((rank, ok))


def test_two_rank_gradient_allreduce():
    """Data-parallel training: per-rank backward on a shard + bucketed all-reduce == the gradient of the whole batch (averaged)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def _grad_worker_uneven(rank, world, port, out_queue):
    """7 utterances over 2 ranks (4 + 3), each rank's loss a MEAN over its own shard (what ctc.training_loss is)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    runner = ShardedForward(world_size=world, rank=rank, device='cpu', backend='gloo')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 40), torch.nn.Tanh(), torch.nn.Linear(40, 5))
    x = torch.randn(7, 7)
    lo, hi = shard_bounds(7, world, rank)
    net(x[lo:hi]).pow(2).sum(dim=1).mean().backward()     # mean over THIS rank's utterances
    plain = [p.grad.clone() for p in net.parameters()]
    runner.allreduce_gradients(list(net.parameters()), n_local=hi - lo)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 40), torch.nn.Tanh(), torch.nn.Linear(40, 5))
    ref.load_state_dict(net.state_dict())
    ref(x).pow(2).sum(dim=1).mean().backward()             # mean over the global batch on one rank
    ok = all(torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6) for p, q in zip(net.parameters(), ref.parameters()))
    # the equal-weight average is NOT that gradient on an uneven split (the bug the weighting fixes)
    for p, g in zip(net.parameters(), plain):
        p.grad.copy_(g)
    runner.allreduce_gradients(list(net.parameters()))
    differs = any(not torch.allclose(p.grad, q.grad, rtol=1e-3, atol=1e-5) for p, q in zip(net.parameters(), ref.parameters()))
    runner.close()
    out_queue.put((rank, ok and differs))


def test_two_rank_gradient_allreduce_uneven_shards_mean_loss():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def test_single_process_is_a_no_op():
    runner = ShardedForward(world_size=1, rank=0, device='cpu')
    x = torch.randn(4, 80, 20)
    assert torch.equal(runner.forward(_fake_model, x), _fake_model(x))
    assert torch.equal(runner.forward_global(_fake_model, x), _fake_model(x))
    assert runner.max_over_ranks(1.5) == 1.5
    runner.barrier()
    runner.close()


# ---- the same two-rank path with the real HIP model: both ranks share the one GPU of the test box, gloo moves the logits ----

def _gpu_worker(rank, world, port, out_queue):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import nb_asr_amd as nb
    from nb_asr_amd.weights import keyed_fill_, keyed_input
    dev = torch.device('cuda', 0)
    runner = ShardedForward(world_size=world, rank=rank, device=dev, backend='gloo')
    model = nb.get_model([[3, 1], [4, 1, 1], [2, 1, 1, 1]], use_rnn=True, dropout_rate=0.0)
    keyed_fill_(model, 1235, 'lively')
    model = model.to(dev).eval()
    global_x = keyed_input(5, 61, seed=3).to(dev)          # 5 utterances over 2 ranks: shards of 3 and 2
    with torch.no_grad():
        want = model(global_x).clone()
        got = runner.forward_global(model, global_x)
        local = runner.forward(model, keyed_input(2, 61, seed=10 + rank).to(dev))
        parts = [model(keyed_input(2, 61, seed=10 + r).to(dev)).clone() for r in range(world)]
    ok_global = torch.equal(got, want)                      # utterances are independent: sharding must not change a bit
    ok_local = torch.equal(local, torch.cat(parts))
    runner.barrier()
    runner.close()
    out_queue.put((rank, ok_global, ok_local))


@pytest.mark.gpu
def test_two_rank_sharded_forward_with_the_hip_model():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok_global, ok_local in results:
        assert ok_global, f'rank {rank}: sharded logits differ from the unsharded forward'
        assert ok_local, f'rank {rank}: all-gathered logits are not the per-rank results in rank order'


def test_single_rank_takes_the_collective_path_when_asked():
    """`force_collective=True`: a single rank initialises a process group and runs the same all-gather / all-reduce / barrier calls as
    N ranks do (what the driver's `torch.distributed.run --nproc-per-node 1` launch exercises with RCCL on the GPU box)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1')
    runner = ShardedForward(world_size=1, rank=0, device='cpu', backend='gloo', force_collective=True)
    try:
        assert runner.collective
        x = torch.randn(3, 80, 20)
        assert torch.equal(runner.forward(_fake_model, x), _fake_model(x))
        assert runner.max_over_ranks(2.5) == 2.5
        p = torch.nn.Parameter(torch.ones(5))
        p.grad = torch.full((5,), 3.0)
        runner.allreduce_gradients([p], n_local=3)
        assert torch.equal(p.grad, torch.full((5,), 3.0))
        runner.barrier()
    finally:
        runner.close()
