"""Multi-GPU execution: batch-sharded replicas, one process per GPU (forward: one all-gather of the logits; training: bucketed all-reduce of the gradients).

Utterances are independent (no BatchNorm, LayerNorm is per frame), so the path shards along the batch with
NO collective inside the forward; every rank holds a full weight replica (105 MB fp32).  The only exchange is
ONE all-gather of the logits shard (batch/G, T', 49) at the end -- RCCL over xGMI when the backend is "nccl"
(RCCL on ROCm), gloo on CPU for tests.  The reference's own multi-GPU code is torch.nn.DataParallel in the
trainer (training/torch/trainer.py:91-92: single process, scatter/replicate/gather); this replaces that pattern
with process-per-GPU replicas.  The shard is sub-megabyte, so the op is latency-bound: a single all-gather
(no bucketing, no overlap) is the right shape for the point-to-point xGMI mesh.
"""
import datetime
import os

import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced [begin, end) of rank's shard (first n_items % world_size ranks get one extra)."""
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class ShardedForward:
    """Owns the process group (if any) and runs ``model`` on this rank's shard."""

    def __init__(self, world_size=None, rank=None, device=None, backend=None, force_collective=False):
        self.world_size = int(os.environ.get('WORLD_SIZE', '1')) if world_size is None else world_size
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else rank
        self.device = device
        self._own_group = False
        # ``force_collective``: take the collective path with a single rank too (tests/test_rccl_gpu.py: a 1-rank nccl group on the
        # GPU box; bench.py --force-collective: every step then ends with the all-gather, `allgather_us` in the line)
        self.collective = self.world_size > 1 or bool(force_collective)
        if self.collective and not dist.is_initialized():
            if backend is None:
                backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            kwargs = {}
            if backend == 'nccl':
                kwargs['device_id'] = torch.device(device)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world_size,
                                    timeout=datetime.timedelta(seconds=600), **kwargs)
            self._own_group = True

    # -- collectives -------------------------------------------------------------------------------------------
    def gather_logits(self, local):
        """All-gather equally sized (b, T', classes) shards into (world*b, T', classes), rank order."""
        if not self.collective:
            return local
        local = local.contiguous()
        out = torch.empty((self.world_size * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out

    def gather_ragged(self, local, n_items):
        """All-gather shards of a global batch of ``n_items`` split by ``shard_bounds`` (sizes may differ by one)."""
        if not self.collective:
            return local
        sizes = [e - b for b, e in (shard_bounds(n_items, self.world_size, r) for r in range(self.world_size))]
        pad = max(sizes)
        buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
        buf[: local.shape[0]] = local
        full = self.gather_logits(buf)
        return torch.cat([full[r * pad: r * pad + sizes[r]] for r in range(self.world_size)], dim=0)

    def forward(self, model, local_x):
        """This rank's utterances through the model, then the one all-gather."""
        return self.gather_logits(model(local_x))

    def forward_global(self, model, global_x):
        """Shard a replicated global batch, forward this rank's slice, gather all logits in original order."""
        begin, end = shard_bounds(global_x.shape[0], self.world_size, self.rank)
        return self.gather_ragged(model(global_x[begin:end]), global_x.shape[0])

    def allreduce_gradients(self, parameters, bucket_bytes=64 << 20, n_local=None):
        """Data-parallel training step (what the reference does with nn.DataParallel, trainer.py:91-92, as one process per GPU): after
        ``loss.backward()`` on this rank's shard, average the gradients over the ranks IN PLACE.  The gradients are flattened into
        buckets of ``bucket_bytes`` (the model's 105 MB of fp32 gradients: two 64 MiB all-reduces -- ring all-reduce over xGMI is bound
        per link, a few large messages amortise its latency where 166 small ones would not) and copied back.  Parameters without a
        gradient on this rank contribute zeros (every rank must pass the same parameter list).

        ``n_local`` = the number of utterances of THIS rank's shard.  Each rank's loss is a mean over its own shard
        (``ctc.training_loss``), so with shards of different sizes (``shard_bounds`` when the batch is not a multiple of the world
        size) the gradient of the GLOBAL-batch mean -- what nn.DataParallel computes from the gathered outputs -- is
        sum_r (n_r / n) grad_r, not the plain average: every rank's gradients are scaled by n_local / n_global (one extra
        all-reduce of the counts) before the SUM.  ``None`` = equal shards (plain average)."""
        params = [p for p in parameters if p.requires_grad]
        if not self.collective or not params:
            return
        if n_local is None:
            weight = 1.0 / self.world_size
        else:
            count = torch.tensor([float(n_local)], dtype=torch.float64, device=params[0].device)
            dist.all_reduce(count, op=dist.ReduceOp.SUM)
            n_global = float(count.item())
            if n_global <= 0:
                raise ValueError('allreduce_gradients: the ranks hold no utterances')
            weight = float(n_local) / n_global
        bucket, size = [], 0

        def flush():
            if not bucket:
                return
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            flat *= weight                                   # n_local / n_global (1 / world for equal shards), then SUM
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            offset = 0
            for p in bucket:
                n = p.numel()
                piece = flat[offset: offset + n].view_as(p)
                if p.grad is None:
                    p.grad = piece.clone()
                else:
                    p.grad.copy_(piece)
                offset += n
            bucket.clear()

        for p in params:
            nbytes = p.numel() * p.element_size()
            if bucket and (size + nbytes > bucket_bytes or p.dtype != bucket[0].dtype):
                flush()
                size = 0
            bucket.append(p)
            size += nbytes
        flush()

    def barrier(self):
        if self.collective:
            dist.barrier()

    def max_over_ranks(self, value):
        if not self.collective:
            return value
        dev = self.device if (self.device is not None and dist.get_backend() == 'nccl') else 'cpu'
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()
