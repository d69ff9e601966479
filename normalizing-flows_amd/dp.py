"""Data-parallel evaluation of the hot path: one process per GPU, batch sharded by rows, weights replicated.

The path is embarrassingly parallel over samples: no data-path collective.  The only
exchange is ONE all-reduce of the fp64 pair [sum log_q, row count] (16 bytes) per evaluated batch for the NLL -- RCCL over
xGMI when the process group backend is "nccl", gloo in the CPU tests.  Training adds the gradient average: bucketed all-reduces
started from autograd hooks while backward is still running (OverlappedGradientAverager), or one call after it
(allreduce_gradients, optionally on ONE persistent flat buffer: FlatGradients).  The reference has no distributed code
(`grep torch.distributed normflows/` is empty); semantics are those of core.py:87-102 `-mean(log_q)` over the
GLOBAL batch.
"""
import torch
import torch.distributed as dist

from . import _sidestream


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


class FlatGradients:
    """ONE persistent flat gradient buffer per dtype; every parameter's `.grad` is a view into it.  autograd accumulates into the
    views in place, `zero()` is one memset instead of one launch per parameter, and `allreduce()` hands the buffer itself to the
    collective (bucket_bytes-sized views of it: no torch.cat, no copy back) -- for the benchmark model 21.8 MB per step that the
    bucketed path below gathers and scatters again.  Keep gradients allocated: `optimizer.zero_grad(set_to_none=False)` or
    `FlatGradients.zero()`; a `.grad` that was replaced (set_to_none=True, or assigned by hand) is re-attached by `attach()`."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.flat = {}
        self.views = []
        groups = {}
        for p in self.params:
            groups.setdefault((p.dtype, p.device), []).append(p)
        for key, ps in groups.items():
            n = sum(p.numel() for p in ps)
            buf = torch.zeros(n, dtype=key[0], device=key[1])
            self.flat[key] = buf
            off = 0
            for p in ps:
                self.views.append((p, buf[off:off + p.numel()].view_as(p)))
                off += p.numel()
        self.attach()

    def attach(self):
        """(Re-)install the views as the parameters' gradients; an existing foreign gradient is copied in first."""
        for p, v in self.views:
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def zero(self):
        self.attach()
        for buf in self.flat.values():
            buf.zero_()

    def allreduce(self, group=None, bucket_bytes=64 << 20):
        """Average over the ranks: all_reduce(SUM) on <= bucket_bytes views of the flat buffers, then one scale.  Returns the
        number of collectives (0 without a process group of more than one rank)."""
        if not (dist.is_initialized() and dist.get_world_size(group) > 1):
            return 0
        self.attach()
        world = dist.get_world_size(group)
        n_coll = 0
        for buf in self.flat.values():
            step = max(1, bucket_bytes // buf.element_size())
            for lo in range(0, buf.numel(), step):
                dist.all_reduce(buf[lo:lo + step], op=dist.ReduceOp.SUM, group=group)
                n_coll += 1
            buf.div_(world)
        return n_coll


class FlatParameters:
    """Every trainable parameter of a module as a VIEW of one flat buffer, and one flat gradient buffer the backward kernels write
    into directly (round 6).

        flat = dp.FlatParameters(model)                       # parameters now alias flat.param; checkpoints are unchanged
        opt = torch.optim.Adam(flat.parameters(), lr=1e-4, fused=True)          # ONE tensor: one launch per step
        for x in batches:
            flat.zero_grad()                                  # host only: the parameters' .grad become None
            loss = model.forward_kld(x); loss.backward()
            flat.sync()                                       # .grad of the flat parameter = the whole model's gradient
            opt.step()

    * Parameters keep their names, shapes and `state_dict` entries; `p.data` is re-pointed to a slice of `self.param` (one
      contiguous buffer in `module.parameters()` order), so `optimizer.step()` on the flat parameter updates them in place.
      torch.optim.Adam(fused=True) over the benchmark model's 608 parameter tensors is 17 multi-tensor launches (0.72 ms per step);
      over one flat tensor one launch.  The packed-weight caches follow steps of the flat parameter (_keys.py: address ranges).
    * Gradients: every parameter gets a view of `self.grad` registered as its gradient DESTINATION (_gradbuf.py).  The one-call
      layer backward (nf_coupling_train_bwd) and LULinearPermute's backward write there directly and autograd adopts those views
      as `p.grad` without a copy; a layer that does not know about destinations leaves an ordinary `p.grad`, which `sync()`
      copies into its slice (and zero-fills the slice of a parameter that received no gradient).  After `sync()`,
      `self.param.grad` holds the model's gradient: one buffer for the optimizer and for the data-parallel all-reduce
      (`OverlappedGradientAverager(..., flat=flat)` / `allreduce()` reduce slices of it in place: no torch.cat, no copy back).
    * One dtype and one device (ValueError otherwise); `.to()` / `.double()` on the module afterwards breaks the aliasing -- build
      the FlatParameters last.  `release()` unregisters the destinations (the parameters stay views of the flat buffer)."""

    def __init__(self, module_or_params):
        from . import _gradbuf
        ps = module_or_params.parameters() if isinstance(module_or_params, torch.nn.Module) else module_or_params
        seen, params = set(), []
        for p in ps:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("FlatParameters: no trainable parameters")
        dt, dev = params[0].dtype, params[0].device
        if any(p.dtype != dt or p.device != dev for p in params):
            raise ValueError("FlatParameters: one dtype and one device per instance")
        n = sum(p.numel() for p in params)
        data = torch.empty(n, dtype=dt, device=dev)
        self.grad = torch.zeros(n, dtype=dt, device=dev)
        self.params, self.views, self.offsets = params, [], []
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                data[off:off + k].copy_(p.detach().reshape(-1))
                p.data = data[off:off + k].view(p.shape)
                v = self.grad[off:off + k].view(p.shape)
                self.views.append((p, v))
                self.offsets.append((off, off + k))
                _gradbuf.register(p, v)
                off += k
        self.param = torch.nn.Parameter(data)
        self.param.grad = self.grad

    def parameters(self):
        return [self.param]

    def zero_grad(self):
        """Host only: the next backward writes every slice (sync() zero-fills what it did not)."""
        for p in self.params:
            p.grad = None
        self.param.grad = self.grad

    def sync(self):
        """Make `self.param.grad` the gradient of the last backward: slices the kernels wrote are already there (their `.grad` IS
        the slice); other gradients are copied in, missing ones zero-filled.  Returns the number of slices that needed a launch."""
        n = 0
        with torch.no_grad():
            for p, v in self.views:
                g = p.grad
                if g is None:
                    v.zero_()
                    n += 1
                elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                    v.copy_(g)
                    n += 1
        self.param.grad = self.grad
        return n

    def allreduce(self, group=None, bucket_bytes=64 << 20):
        """sync(), then average the flat gradient over the ranks in place (<= bucket_bytes slices).  Returns the collectives."""
        self.sync()
        if not (dist.is_initialized() and dist.get_world_size(group) > 1):
            return 0
        world = dist.get_world_size(group)
        step = max(1, bucket_bytes // self.grad.element_size())
        n_coll = 0
        for lo in range(0, self.grad.numel(), step):
            dist.all_reduce(self.grad[lo:lo + step], op=dist.ReduceOp.SUM, group=group)
            n_coll += 1
        self.grad.div_(world)
        return n_coll

    def release(self):
        from . import _gradbuf
        for p in self.params:
            _gradbuf.release(p)


class OverlappedGradientAverager:
    """Gradient averaging OVERLAPPED with the backward pass: parameters are grouped into buckets in reverse registration order
    (the order loss.backward() finishes them); a post-accumulate hook counts a bucket's parameters down, and once the last one
    has its gradient the bucket is flattened and its all_reduce(SUM) is started asynchronously (RCCL runs it on its own stream
    over xGMI while the remaining layers' backward kernels keep the compute queue busy).  `finish()` after backward waits for
    the collectives, scales by 1 / world_size and scatters the averages back.  Same result as allreduce_gradients.

        avg = dp.OverlappedGradientAverager(model.parameters(), bucket_bytes=8 << 20)
        loss.backward(); avg.finish(); optimizer.step()

    Contract:
      * collectives are issued in BUCKET-INDEX order on every rank, whatever order the hooks fire in: a completed bucket waits
        for its predecessors, and what backward() did not complete is issued by finish(), still in index order, every bucket at
        its FULL size (a parameter without a gradient contributes zeros and gets nothing written back) -- so ranks on which
        different parameters went unused still run the same sequence of equally sized all_reduces;
      * one backward per finish(): a second backward() before finish() raises instead of silently averaging the first
        backward's gradients over the accumulated ones.  Gradient accumulation: run the extra backward passes under
        `with avg.no_sync():` (hooks off, gradients accumulate locally) and the last one outside it.

    Without a process group of more than one rank the hooks do nothing."""

    def __init__(self, params, group=None, bucket_bytes=8 << 20, flat=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self._flat_init(flat)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat_ranges()
        self._sync = True
        self._reset()
        self._ready_fn = _sidestream.hook_is_aware(self._ready)     # one bound-method object for every parameter: it joins the side stream itself
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready_fn) for p in self.params]

    def _reset(self):
        self._left = [len(b) for b in self.buckets]
        self._next = 0              # the next bucket index to issue
        self._pending = []          # (bucket index, flat tensor, work handle), in issue (= index) order

    def _active(self):
        return self._sync and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def no_sync(self):
        """Context manager: backward passes inside it only accumulate into .grad (no collective, nothing counted)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self._pending or self._next:
                raise RuntimeError("OverlappedGradientAverager.no_sync(): collectives of an unfinished step are in flight; call finish() first")
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old
        return ctx()

    # ---- flat=: reduce slices of a persistent flat gradient buffer IN PLACE (FlatGradients / FlatParameters; round 6) ----------
    # A bucket is a run of parameters that are consecutive in registration order, i.e. one contiguous slice [lo, hi) of the flat
    # buffer when the buffer was built from the same parameter list: the collective runs on that slice -- no torch.cat in front of
    # it, no copy back behind it (2 x 21.8 MB of traffic per step of the benchmark model on the gather / scatter path).
    def _flat_init(self, flat):
        self._where = None
        if flat is None:
            return
        self._where = {}
        if isinstance(flat, FlatParameters):
            for (p, v), (lo, hi) in zip(flat.views, flat.offsets):
                self._where[id(p)] = (flat.grad, lo, hi, v)
        else:       # FlatGradients: one buffer per (dtype, device), parameters in the order given
            offs = {}
            for p, v in flat.views:
                key = (p.dtype, p.device)
                lo = offs.get(key, 0)
                self._where[id(p)] = (flat.flat[key], lo, lo + p.numel(), v)
                offs[key] = lo + p.numel()

    def _flat_ranges(self):
        self._ranges = None
        if self._where is None:
            return
        self._ranges = []
        for b in self.buckets:
            if any(id(q) not in self._where for q in b):
                raise ValueError("OverlappedGradientAverager(flat=...): a parameter is not part of the flat buffer")
            buf = self._where[id(b[0])][0]
            lo, hi = min(self._where[id(q)][1] for q in b), max(self._where[id(q)][2] for q in b)
            if any(self._where[id(q)][0] is not buf for q in b) or hi - lo != sum(q.numel() for q in b):
                raise ValueError("OverlappedGradientAverager(flat=...): build the flat buffer from the same parameter list (a bucket "
                                 "must be one contiguous slice of it)")
            self._ranges.append((buf, lo, hi))

    def _issue(self, i):
        if self._ranges is not None:
            buf, lo, hi = self._ranges[i]
            for q in self.buckets[i]:         # gradients the kernels / autograd did not leave in the slice itself
                v, g = self._where[id(q)][3], q.grad
                if g is None:
                    v.zero_()                 # (full-size contract: a parameter without a gradient contributes zeros)
                elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                    v.copy_(g)
            flat = buf[lo:hi]
        else:
            flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in self.buckets[i]])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((i, flat, work))

    def _ready(self, p):
        if not self._active():
            return
        _sidestream.join()          # the pair backward's reduction launches may still be on the side stream (_sidestream.py)
        i = self._bucket_of[id(p)]
        if self._left[i] <= 0 or i < self._next:
            raise RuntimeError("OverlappedGradientAverager: a second backward() reached a bucket whose all_reduce was already started; "
                               "call finish() after every backward(), or accumulate under `with averager.no_sync():`")
        self._left[i] -= 1
        while self._next < len(self.buckets) and self._left[self._next] == 0:     # index order on every rank
            self._issue(self._next)
            self._next += 1

    def finish(self):
        """Issue (in index order, full size) the buckets backward() did not complete, wait for everything, write the averages
        back into the gradients that exist.  Returns the number of collectives (= the number of buckets)."""
        if not self._active():
            return 0
        world = dist.get_world_size(self.group)
        late = len(self.buckets) - self._next
        while self._next < len(self.buckets):
            self._issue(self._next)
            self._next += 1
        if late == len(self.buckets) and late > 1 and not getattr(self, "_warned_late", False):
            #  nothing overlapped: bucket 0 never completed during backward -- typically an unused parameter in it
            import warnings
            warnings.warn("OverlappedGradientAverager.finish(): no bucket completed during backward (a parameter of the first "
                          "bucket got no gradient?): every all_reduce was issued here, nothing overlapped with the backward pass")
            self._warned_late = True
        n = len(self._pending)
        for i, flat, work in self._pending:
            work.wait()
            flat.div_(world)
            if self._ranges is not None:
                for q in self.buckets[i]:     # in place: only a gradient that lives elsewhere gets its average copied back
                    v, g = self._where[id(q)][3], q.grad
                    if g is not None and (g.data_ptr() != v.data_ptr() or g.stride() != v.stride()):
                        g.copy_(v)
                continue
            off = 0
            for q in self.buckets[i]:
                if q.grad is not None:
                    q.grad.copy_(flat[off:off + q.numel()].view_as(q.grad))
                off += q.numel()
        self._reset()
        return n

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        _sidestream._aware.discard(id(self._ready_fn))


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Average the gradients of `params` over all ranks: flat fp32/fp64 buckets (<= bucket_bytes each, sized for
    per-link-bound ring collectives on point-to-point xGMI: few large messages), one all_reduce(SUM) per bucket, then a
    scale by 1 / world_size.  Same math as the reference's single-process `-mean(log_q)` over the global batch when
    every rank holds an equal share of the rows.  No-op without an initialised process group.  `params` may be a FlatGradients
    (gradients as views of one persistent buffer: the collective runs on the buffer itself); for plain parameter lists the
    buckets are gathered with torch.cat and copied back."""
    if isinstance(params, FlatGradients):
        return params.allreduce(group=group, bucket_bytes=bucket_bytes)
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
