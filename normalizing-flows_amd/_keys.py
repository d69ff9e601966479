"""Cache keys of the packed-weight images (fused blobs, one-launch packs, composed LU matrices, masked weights).

An image is valid while the parameters it was built from are unchanged.  `(data_ptr, _version)` sees every in-place update made
through the parameter itself -- but NOT the updates of torch's fused optimizers (`torch.optim.Adam(..., fused=True)` and friends go
through `torch._fused_*_`, which leaves `Tensor._version` untouched on CPU and GPU alike) and not `.data` updates.  So every key
also carries, per tensor, the number of the last optimizer step that had this tensor in its `param_groups` (a global optimizer
post-step hook records it by `data_ptr`): after an `optimizer.step()` the images built from THAT optimizer's parameters are
rebuilt at their next use, whichever implementation the optimizer chose -- a frozen / EMA / teacher model in the same process keeps
its packs and its recorded hipGraphs (round 4 advanced one process-wide epoch on any optimizer's step: ADVICE r04, _keys.py:31).
`.data` updates stay the caller's business (`normflows_amd.invalidate_caches` -> `bump()`: the process-wide epoch, part of every
key)."""
_epoch = 0          # advanced by bump() only: invalidates everything
_steps = 0          # optimizer steps seen so far (any optimizer)
_stepped = {}       # data_ptr -> number of the last optimizer step that updated the tensor at that address


def bump():
    """Advance the epoch: every packed-weight cache keyed with `pkey` is stale from here on."""
    global _epoch
    _epoch += 1


def epoch():
    return _epoch


def stamp():
    """Changes whenever ANYTHING may have been invalidated (an optimizer step or bump()): the cheap first test of holders of
    derived state (core._GraphCache), which then look at their own parameters with `signature`."""
    return (_epoch, _steps)


def signature(tensors):
    """(epoch, latest optimizer step that touched any of `tensors`)."""
    last = 0
    for t in tensors:
        s = _stepped.get(t.data_ptr(), 0)
        if s > last:
            last = s
    return (_epoch, last)


def pkey(tensors):
    return tuple((t.data_ptr(), t._version, _stepped.get(t.data_ptr(), 0)) for t in tensors) + (_epoch,)


def _after_step(optimizer, args, kwargs):
    global _steps
    _steps += 1
    for group in optimizer.param_groups:
        for p in group["params"]:
            _stepped[p.data_ptr()] = _steps


def _install():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:      # (older torch: the (data_ptr, _version) part of the key is all there is)
        return
    register_optimizer_step_post_hook(_after_step)


_install()
