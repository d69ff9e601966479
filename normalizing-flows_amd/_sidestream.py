"""The side stream of the training step (round 6, late).

The backward of a benchmark-shaped [CoupledRQS, LULinearPermute] pair ends with two launches that only produce PARAMETER gradients --
the one reduction of the pair's partial tiles and the LU's factor gradients, 46 us of latency-bound work per pair -- while the next
pair's backward (the previous pair of the model) only waits for the INPUT gradient.  autograd.PairTrainFn issues those two launches on
the stream kept here, forked from the current stream by an event (ops.pair_train_bwd(side=...): nf_pair_train_bwd_head / _tail), so they
can run under the next pair's kernels.  OPT-IN (config.set_train_reduce_async(True)): measured, it buys 0.2-0.3 ms of a 25.3 ms step at
best and nothing since the reduction launch schedules its longest blocks first -- the heavy kernels allocate a CU's whole register file,
a second kernel only gets the drain / ramp at their boundaries (config.py).

Who joins.  Nothing may read such a gradient on another stream before `join()`:
  * the end of the backward pass: PairTrainFn queues `join` on the autograd engine (queue_callback) -- whatever follows
    `loss.backward()` / `torch.autograd.grad()` on the current stream (optimizer step, clipping, all-reduce, .grad reads) is ordered
    behind the side stream's work; under hipGraph capture the same callback closes the fork inside the capture;
  * dp.OverlappedGradientAverager calls `join()` before it starts a bucket's all-reduce during backward.
PairTrainFn takes the side stream only when no other reader can exist before that: every parameter of the pair writes into a registered
gradient buffer whose .grad is unset (autograd adopts the returned view: no accumulation kernel on the current stream) and has no
tensor hooks (post-accumulate hooks declared join-aware excepted: hook_is_aware).  
"""
import torch

_streams = {}         # device index -> torch.cuda.Stream
_dirty = set()        # device indices with side work not yet joined
_keep = {}            # device index -> tensors the side work reads or writes, kept alive until the join (see fork)
_aware = set()        # id() of post-accumulate-grad hook callables that call join() themselves before they read a gradient


def stream(device):
    """The side stream of `device` (created outside any capture: None while the current stream is capturing and none exists yet)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(idx)
    if s is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        s = _streams[idx] = torch.cuda.Stream(device=idx)
    return s


def hook_is_aware(fn):
    """Declare a post-accumulate-grad hook callable (the very object passed to register_post_accumulate_grad_hook) as joining by
    itself; returns fn."""
    _aware.add(id(fn))
    return fn


def nobody_reads_early(param):
    """True when autograd will adopt a gradient written for `param` without running anything that reads it: no .grad to accumulate
    into, no tensor hooks, and only join-aware post-accumulate hooks."""
    if param.grad is not None or param._backward_hooks:
        return False
    post = getattr(param, "_post_accumulate_grad_hooks", None)
    return not post or all(id(h) in _aware for h in post.values())


def mark(device):
    _dirty.add(device.index if device.index is not None else torch.cuda.current_device())


def join():
    """Order the current stream of every device with pending side work behind it.  Cheap when nothing is pending."""
    while _dirty:
        idx = _dirty.pop()
        torch.cuda.current_stream(idx).wait_stream(_streams[idx])
        _keep.pop(idx, None)      # released only now: whatever reuses their memory on the current stream is ordered behind the wait


def fork(device, keep=()):
    """Inside a backward function: the side stream, ordered behind everything the current stream has been given so far, for launches
    that only produce PARAMETER gradients (Glow's leaves, round 6: a GlowBlock's backward hands the previous block an input gradient
    after ~60 % of its launches; the conditioner's weight gradients, the 1x1 convolution's weight / LU-factor gradients and their
    reductions feed nothing downstream, and at the 8x8 / 4x4 levels no kernel of the step fills the chip).  `keep`: every tensor
    the side launches touch whose last Python reference may die before the join -- the caching allocator would hand their memory to
    the next allocation on the CURRENT stream while the side stream still reads it; they are released by join().  Tensors allocated
    while the side stream is current belong to its pool and need no entry.  The join is queued on the autograd engine.  None when
    no side stream may be created (first use under capture)."""
    s = stream(device)
    if s is None:
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s.wait_stream(torch.cuda.current_stream(idx))
    _keep.setdefault(idx, []).extend(keep)
    _dirty.add(idx)
    queue_join()
    return s


def queue_join():
    """Inside a backward function: run join() when this backward pass ends.  Queued by every caller (a pass that died with an exception
    never ran its callbacks: a once-per-pass flag would be left set); the second and later calls of a pass find nothing pending."""
    torch.autograd.Variable._execution_engine.queue_callback(join)
