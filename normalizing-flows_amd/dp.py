"""Data-parallel evaluation of the hot path: one process per GPU, batch sharded by rows, weights replicated.

The path is embarrassingly parallel over samples: no data-path collective.  The only
exchange is ONE all-reduce of the fp64 pair [sum log_q, row count] (16 bytes) per evaluated batch for the NLL -- RCCL over
xGMI when the process group backend is "nccl", gloo in the CPU tests.  The reference has no distributed code
(`grep torch.distributed normflows/` is empty); semantics are those of core.py:87-102 `-mean(log_q)` over the
GLOBAL batch.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous row range [lo, hi) of `rank`: rows are split as evenly as possible, low ranks get the extras."""
    base, extra = divmod(n_rows, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_rows(x, world_size=None, rank=None):
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def global_nll(log_q_local, group=None):
    """-mean(log_q) over all ranks' rows: per call ONE all_reduce(SUM) of the fp64 pair [sum log_q, row count] (16 bytes).
    The count travels with the sum on every call -- shards may be uneven and may change from call to call, so nothing is
    cached (a cached count keyed by the local row count would desynchronise the ranks' collectives)."""
    s = log_q_local.sum(dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        packed = torch.stack([s, torch.full((), float(log_q_local.numel()), dtype=torch.float64, device=s.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        return -packed[0] / packed[1]
    return s * (-1.0 / max(log_q_local.numel(), 1))


def sharded_forward_kld(log_prob_fn, x_local, group=None):
    """forward_kld of core.py:87-102 on a row-sharded batch: local log_prob, then the single NLL all-reduce."""
    return global_nll(log_prob_fn(x_local), group=group)


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Average the gradients of `params` over all ranks: flat fp32/fp64 buckets (<= bucket_bytes each, sized for
    per-link-bound ring collectives on point-to-point xGMI: few large messages), one all_reduce(SUM) per bucket, then a
    scale by 1 / world_size.  Same math as the reference's single-process `-mean(log_q)` over the global batch when
    every rank holds an equal share of the rows.  No-op without an initialised process group."""
    params = [p for p in params if p.grad is not None]
    if not (dist.is_initialized() and dist.get_world_size(group) > 1) or not params:
        return 0
    world = dist.get_world_size(group)
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.grad.numel() * p.grad.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or p.grad.dtype != cur[0].grad.dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return len(buckets)


def combine_moments(mean, std, count, group=None):
    """Per-channel mean and UNBIASED std of the GLOBAL batch from every rank's local (mean, std, count): one
    all_reduce(SUM) of 2C + 1 fp64 numbers.  Used by ActNorm's data-dependent initialisation (normalization.py:19-39)
    so that N data-parallel replicas initialise to exactly the parameters a single process would compute on the whole
    batch -- with shard-local statistics the replicas' weights would silently differ."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return mean, std
    n = float(count)
    m64, s64 = mean.double(), std.double()
    packed = torch.cat([n * m64, (n - 1.0) * s64 * s64 + n * m64 * m64,
                        torch.tensor([n], dtype=torch.float64, device=mean.device)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    C = mean.numel()
    N = packed[-1]
    gmean = packed[:C] / N
    gvar = (packed[C:2 * C] - N * gmean * gmean) / (N - 1.0)
    return gmean.to(mean.dtype), gvar.clamp_min(0).sqrt().to(std.dtype)
