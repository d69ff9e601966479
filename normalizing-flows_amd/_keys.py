"""Cache keys of the packed-weight images (fused blobs, one-launch packs, composed LU matrices, masked weights).

An image is valid while the parameters it was built from are unchanged.  `(data_ptr, _version)` sees every in-place update made
through the parameter itself -- but NOT the updates of torch's fused optimizers (`torch.optim.Adam(..., fused=True)` and friends go
through `torch._fused_*_`, which leaves `Tensor._version` untouched on CPU and GPU alike) and not `.data` updates.  So every key
also carries a process-wide EPOCH that a global optimizer post-step hook advances: after any `optimizer.step()` every image is
rebuilt at its next use, whichever implementation the optimizer chose.  `.data` updates stay the caller's business
(`normflows_amd.invalidate_caches`)."""
_epoch = 0


def bump():
    """Advance the epoch: every packed-weight cache keyed with `pkey` is stale from here on."""
    global _epoch
    _epoch += 1


def epoch():
    return _epoch


def pkey(tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors) + (_epoch,)


def _install():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:      # (older torch: the (data_ptr, _version) part of the key is all there is)
        return
    register_optimizer_step_post_hook(lambda optimizer, args, kwargs: bump())


_install()
