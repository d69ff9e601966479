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
_epoch = 0          # advanced by bump(): invalidates everything
_steps = 0          # optimizer steps seen so far (any optimizer)
_stepped = {}       # data_ptr -> number of the last optimizer step that updated the tensor at that address
_seen = set()       # data_ptrs of the tensors cache keys were built from (pkey / signature)
_ranges = []        # [lo, hi, step]: stepped tensors that are NOT key tensors themselves but contain key tensors (flat buffers)

# Round 6.  An optimizer may step tensors that are not the modules' own parameters: ONE flat parameter whose slices the
# parameters are views of (dp.FlatParameters), master-weight copies written back through `.data.copy_`, a ZeRO-style wrapper.
#   * a stepped tensor whose address range CONTAINS key tensors is recorded as a range: every key tensor inside it counts as
#     stepped (precise: other models keep their packs);
#   * a stepped tensor that is neither a key tensor nor contains one has an unknown relation to the cached images: the
#     process-wide epoch advances (coarse, safe -- round 4's behaviour for exactly these cases).
# Both tables are pruned when they grow (a stale entry can only cause a rebuild, never a missed one: step numbers only increase).
_PRUNE = 1 << 16


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


def _last_step(ptr):
    s = _stepped.get(ptr, 0)
    for lo, hi, st in _ranges:
        if lo <= ptr < hi and st > s:
            s = st
    return s


def _note(ptr):
    if ptr not in _seen:
        if len(_seen) >= 16 * _PRUNE:       # (never in practice: forget everything, conservatively)
            _seen.clear()
            bump()
        _seen.add(ptr)


def signature(tensors):
    """(epoch, latest optimizer step that touched any of `tensors`)."""
    last = 0
    for t in tensors:
        ptr = t.data_ptr()
        _note(ptr)
        s = _last_step(ptr)
        if s > last:
            last = s
    return (_epoch, last)


def pkey(tensors):
    out = []
    for t in tensors:
        ptr = t.data_ptr()
        _note(ptr)
        out.append((ptr, t._version, _last_step(ptr)))
    return tuple(out) + (_epoch,)


_class = {}         # (lo, hi) of a stepped tensor -> (len(_seen) when classified, kind): 0 plain parameter, 1 range, 2 unknown


def _classify(lo, hi):
    ent = _class.get((lo, hi))
    if ent is not None and ent[0] == len(_seen):
        return ent[1]
    if len(_class) > _PRUNE:
        _class.clear()
    inside = any(lo < q < hi for q in _seen)
    kind = 1 if inside else (0 if lo in _seen else 2)
    _class[(lo, hi)] = (len(_seen), kind)
    return kind


def _after_step(optimizer, args, kwargs):
    global _steps, _stepped
    _steps += 1
    unknown = False
    for group in optimizer.param_groups:
        for p in group["params"]:
            lo = p.data_ptr()
            hi = lo + p.numel() * p.element_size()
            kind = _classify(lo, hi)
            if kind == 0:
                _stepped[lo] = _steps         # an ordinary parameter
            elif kind == 1:                   # a flat buffer over key tensors
                if lo in _seen:
                    _stepped[lo] = _steps
                for r in _ranges:
                    if r[0] == lo and r[1] == hi:
                        r[2] = _steps
                        break
                else:
                    _ranges.append([lo, hi, _steps])
            else:
                unknown = True                # master copies / foreign tensors: what they write back to is not visible from here
    if unknown:
        bump()
    if len(_stepped) > _PRUNE:
        _stepped = {k: v for k, v in _stepped.items() if k in _seen}
    if len(_ranges) > 64:
        del _ranges[:-64]
        bump()


def _install():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:      # (older torch: the (data_ptr, _version) part of the key is all there is)
        return
    register_optimizer_step_post_hook(_after_step)


_install()
