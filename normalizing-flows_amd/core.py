"""Model containers: the callers of the hot path.

Mirrors normflows/core.py:9-213 (NormalizingFlow) and :455-655 (MultiscaleFlow): same constructor
signatures, method names, return values and state_dict layout (`q0.*`, `flows.{i}.*`, `merges.*`), so
reference checkpoints load with strict=True and user code written against nf.NormalizingFlow keeps working.

Differences that are deliberate (MI355X-first):
  * the per-layer `log_q += log_det` is folded into each layer's kernel through the `_run` accumulate
    protocol (flows/base.py) instead of one extra elementwise launch per layer;
  * `use_graphs(True)` records a whole log_prob / sample pass (64+ launches for the 32-layer RQ-NSF) into a
    hipGraph per batch shape and replays it, removing the per-launch host overhead;
  * training runs the same HIP forward kernels inside torch.autograd.Functions (autograd.py); recorded graphs are
    bypassed whenever gradients are required.
"""
import torch
from . import _keys
from . import config as _config
from torch import nn

from . import _prepack
from .flows.base import run_flow
from .flows.mixing import LULinearPermute
from .flows.neural_spline import CoupledRationalQuadraticSpline


def invalidate_caches(module):
    """Drop every packed-weight cache under `module` (fused blobs, one-launch packs, composed LU matrices, masked weights, recorded
    graphs).  The caches are keyed by (data_ptr, _version) of the parameters they were built from, which every in-place update
    through the parameter itself bumps (load_state_dict, p.add_(...), p.copy_(...), the for-loop / foreach optimizers), plus a
    process-wide epoch that a global optimizer post-step hook advances (_keys.py: torch's FUSED optimizers do not bump
    `_version`).  Updates through `.data`
    (`p.data.add_(...)`, `p.data.copy_(...)`, hand-written SGD / EMA on .data) do NOT bump `_version`: call this afterwards, or
    update `p` under torch.no_grad() instead."""
    for m in module.modules():
        for k in list(m.__dict__.keys()):
            if k.endswith("_cache") and not k.startswith("__"):
                v = m.__dict__[k]
                m.__dict__[k] = {} if isinstance(v, dict) else None
        if hasattr(m, "refresh_graphs"):
            m.refresh_graphs()
    _realnvp_cache.clear()


def _pair_signature(crqs):
    p = crqs.prqct
    return (p.features, p.transform_net.hidden_features, len(p.transform_net.blocks), p.num_bins, p.tail_bound,
            p.min_bin_width, p.min_bin_height, p.min_derivative)


def _run_pairs(pairs, z, inverse, ld, acc):
    """`pairs` = [(CoupledRQS, LULinearPermute), ...] in processing order, all of one shape: one persistent launch
    (exact-fp32 kernel nf_rqs_fused_chain, or its split-bf16 counterpart nf_rqs_fused_x3_chain)."""
    from . import _prof
    with _prof.range_("rqs_fused_chain[%d x (CoupledRQS + LULinearPermute)].%s" % (len(pairs), "inverse" if inverse else "forward")):
        return _run_pairs_impl(pairs, z, inverse, ld, acc)


def _run_pairs_impl(pairs, z, inverse, ld, acc):
    from . import config, ops
    if (config.fused_gemm not in ("f32", "bf16x3") or len(pairs) == 1 or not config.fused_chain
            or not pairs[0][0].prqct._fused_eligible(z, None)):       # (pairs beyond the benchmark kernel's shapes: nf_nsf_wide per pair)
        for c, lu in pairs:
            z = c._run_pair(z, lu, inverse, ld, acc)
        return z
    x3 = config.fused_gemm == "bf16x3" and pairs[0][0].prqct.num_bins == 8
    feats, hidden, nblk, K, tb, mw, mh, md = _pair_signature(pairs[0][0])
    from .flows.neural_spline import FUSED_D, FUSED_H
    narrow = z.shape[1] != FUSED_D
    if narrow:        # narrower layers ride the kernel's 64 columns; the padding sits in the splines' tails (log-det 0)
        z = pairs[0][0].prqct._pad_rows(z)
    for i in range(0, len(pairs), 64):
        chunk = pairs[i:i + 64]
        blobs = [(c.prqct._fused_x3_blob(lu) if x3 else c.prqct._fused_blob(lu)) for c, lu in chunk]
        pars = [c.prqct._fused_parity for c, lu in chunk]
        run = ops.rqs_fused_x3_chain if x3 else ops.rqs_fused_chain
        z, _ = run(z, blobs, pars, FUSED_H if x3 else pairs[0][0].prqct._fused_hidden(), nblk, K, 0 if inverse else 1,
                   logdet=ld, acc=acc, tail_bound=tb, live_d=pairs[0][0].prqct._fused_live_d(),
                   min_bin_width=mw, min_bin_height=mh, min_derivative=md, fuse_lu=True)
    return z[:, :feats].contiguous() if narrow else z


# ---- RealNVP-style runs: MaskedAffineFlow(MLP, MLP) / ActNorm stacks as one launch (csrc/realnvp_chain.hip) --------------
def _plain_mlp(net):
    """(linears, slope) if `net` is a nets.MLP made of Linear + LeakyReLU only (mlp.py:5-58 without output_fn/dropout)."""
    from . import nets
    if not isinstance(net, nets.MLP):
        return None
    lin, slope = [], 0.0
    mods = list(net.net)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            lin.append(m)
        elif isinstance(m, nn.LeakyReLU):
            if i == 0 or not isinstance(mods[i - 1], nn.Linear) or i == len(mods) - 1:
                return None
            slope = float(m.negative_slope)
        else:
            return None
    if not lin or any(isinstance(mods[i], nn.Linear) and isinstance(mods[i + 1], nn.Linear) for i in range(len(mods) - 1)):
        return None
    return lin, slope


def _realnvp_record_ok(f, d):
    """True when flow `f` can be a record of the RealNVP chain kernel for (B, d) inputs."""
    from .flows.affine import AffineConstFlow, MaskedAffineFlow
    from .flows.normalization import ActNorm
    if isinstance(f, MaskedAffineFlow):
        if f.b.numel() != d:
            return False
        for net in (f.s, f.t):
            if net is None:
                continue
            pm = _plain_mlp(net)
            if pm is None or pm[0][0].in_features != d or pm[0][-1].out_features != d:
                return False
            if any(l.out_features > 64 or l.bias is None for l in pm[0]):
                return False
        return True
    if isinstance(f, AffineConstFlow):
        if f.s.numel() != d or f.t.numel() != d:
            return False
        if isinstance(f, ActNorm):
            if f._init_known is None:
                f._init_known = bool(f.data_dep_init_done.item() > 0.0)
            return f._init_known          # the data-dependent initialisation needs the layer-by-layer path once
        return True
    return False


def _pack_realnvp(run, d, device):
    """Blob of csrc/realnvp_chain.hip for the flows `run` (stored in forward order)."""
    from .flows.affine import MaskedAffineFlow
    recs, hmax = [], d
    for f in run:
        if isinstance(f, MaskedAffineFlow):
            r = [torch.tensor([2.0, float(f.s is not None), float(f.t is not None), 0.0]), f.b.detach().reshape(-1).float().cpu()]
            for net in (f.s, f.t):
                if net is None:
                    continue
                lin, slope = _plain_mlp(net)
                sizes = [lin[0].in_features] + [l.out_features for l in lin]
                hmax = max(hmax, max(sizes))
                r.append(torch.tensor([float(len(lin)), slope] + [float(v) for v in sizes]))
                for l in lin:
                    r += [l.weight.detach().reshape(-1).float().cpu(), l.bias.detach().reshape(-1).float().cpu()]
            recs.append(torch.cat(r))
        else:
            recs.append(torch.cat([torch.tensor([1.0, 0.0, 0.0, 0.0]), f.s.detach().reshape(-1).float().cpu(),
                                   f.t.detach().reshape(-1).float().cpu()]))
    offs, o = [], 4 + len(recs)
    for r in recs:
        offs.append(float(o))
        o += r.numel()
    blob = torch.cat([torch.tensor([float(len(recs)), float(d), float(hmax), 0.0] + offs)] + recs)
    return blob.to(device), hmax


def _run_realnvp(run, cache, z, inverse, ld, acc):
    from . import ops
    d = z.shape[1]
    key = _keys.pkey(p for f in run for p in list(f.parameters()) + list(f.buffers())) + (str(z.device),)
    ent = cache.get(id(run[0]))
    if ent is None or ent[0] != key:
        ent = (key,) + _pack_realnvp(run, d, z.device)
        cache[id(run[0])] = ent
    y, _ = ops.realnvp_chain(z, ent[1], d, ent[2], 1 if inverse else 0, logdet=ld, acc=acc)
    return y


_realnvp_cache = {}


class _frozen_parameters:
    """requires_grad switched off on every parameter of `module` for the `with` body and ALWAYS restored (a layer without a
    differentiable path raises mid-loop: the model must not stay frozen).  reverse_kld(score_fn=False), core.py:353-363."""

    def __init__(self, module):
        self.params = list(module.parameters())

    def __enter__(self):
        self.req = [p_.requires_grad for p_ in self.params]
        for p_ in self.params:
            p_.requires_grad_(False)

    def __exit__(self, *exc):
        for p_, r in zip(self.params, self.req):
            p_.requires_grad_(r)
        return False


def run_chain(flows, z, inverse, ld, acc):
    """_run_chain_impl behind the training step's one-launch weight packing (_prepack.py): a differentiable density pass first
    packs every eligible layer's weights / LU factors with one launch per kind; the layers then skip their own pack launches."""
    token = _prepack.begin(flows, z, inverse)
    try:
        return _run_chain_impl(flows, z, inverse, ld, acc)
    finally:
        _prepack.end(token)


def _run_chain_impl(flows, z, inverse, ld, acc):
    """Run a list of flows in order (inverse=False) or reversed with .inverse (inverse=True), folding log-dets into
    `ld`.  Adjacent [CoupledRationalQuadraticSpline, LULinearPermute] pairs of the supported shape are fused
    (csrc/rqs_fused.hip) and consecutive fused pairs of one shape run as ONE persistent launch; everything else goes
    layer by layer."""
    n = len(flows)
    order = list(range(n - 1, -1, -1)) if inverse else list(range(n))
    k = 0
    pending = []  # consecutive fusable pairs of one shape

    def flush(zz):
        if pending:
            zz = _run_pairs(list(pending), zz, inverse, ld, acc)
            pending.clear()
        return zz

    rn_ok = (z.dim() == 2 and z.dtype == torch.float32 and z.is_cuda and z.shape[1] <= 16
             and not (torch.is_grad_enabled() and (z.requires_grad
                                                   or any(p.requires_grad for f_ in flows for p in f_.parameters()))))
    while k < n:
        i = order[k]
        f = flows[i]
        if rn_ok and _realnvp_record_ok(f, z.shape[1]):
            k2 = k
            while k2 < n and _realnvp_record_ok(flows[order[k2]], z.shape[1]):
                k2 += 1
            if k2 - k >= 2:   # a run of at least two layers: one launch
                z = flush(z)
                idx = sorted(order[k:k2])
                z = _run_realnvp([flows[q] for q in idx], _realnvp_cache, z, inverse, ld, acc)
                k = k2
                continue
        pair = None
        if k + 1 < n:
            j = order[k + 1]
            a, b = (flows[j], f) if inverse else (f, flows[j])   # a = CoupledRQS candidate, b = LU candidate
            if (inverse and isinstance(a, CoupledRationalQuadraticSpline) and isinstance(b, LULinearPermute)
                    and _prepack.take_pair(a.prqct, b)):
                # training step (round 6): this step's multi-layer packs cover the pair -> one forward launch, composed LU backward
                z = flush(z)
                z = a._run_pair_train(z, b, ld, acc)
                k += 2
                continue
            if (isinstance(a, CoupledRationalQuadraticSpline) and isinstance(b, LULinearPermute)
                    and a._pair_eligible(z, b)):
                pair = (a, b)
        if pair is not None:
            if pending and _pair_signature(pending[0][0]) != _pair_signature(pair[0]):
                z = flush(z)
            pending.append(pair)
            k += 2
        else:
            z = flush(z)
            z = run_flow(f, z, inverse, ld, acc)
            k += 1
    return flush(z)


def _cached_tensors(obj, out, depth=0):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (tuple, list)) and depth < 4:
        for o in obj:
            _cached_tensors(o, out, depth + 1)
    elif isinstance(obj, dict) and depth < 4:
        for o in obj.values():
            _cached_tensors(o, out, depth + 1)


def _packed_blobs(owner):
    """Every tensor held by a `*_cache` attribute of the modules under `owner` (packed weight blobs keyed by parameter
    version) and by the RealNVP chain cache: what a recorded graph has baked pointers to."""
    out = []
    if owner is not None:
        for m in owner.modules():
            for k, v in m.__dict__.items():
                if k.endswith("_cache"):
                    _cached_tensors(v, out)
    _cached_tensors(list(_realnvp_cache.values()), out)
    return out


class _GraphCache:
    """hipGraph replay of a fixed-shape pass.  Inputs are copied into a static buffer; outputs are static.  The entry
    keeps the packed weight blobs it was captured with alive, so a replay after a repack reads stale -- never freed --
    memory (parameters are meant to be frozen while graphs are on; refresh_graphs() re-captures)."""

    def __init__(self, owner=None):
        self.enabled = False
        self.graphs = {}
        self.stamp, self.sig = _keys.stamp(), None
        self.owner = None if owner is None else __import__("weakref").ref(owner)

    def clear(self):
        self.graphs = {}

    def run(self, key, fn, *inputs):
        if not self.enabled:
            return fn(*inputs)
        if self.sig is None or self.stamp != _keys.stamp():
            # some optimizer stepped (or caches were invalidated) since we last looked: were OUR parameters among the updated ones?
            # Then the graphs hold the old packed weights.  (A step of another model's optimizer costs this one scan, not a
            # re-capture: ADVICE r04.)
            owner = self.owner() if self.owner is not None else None
            sig = _keys.signature(owner.parameters()) if owner is not None else _keys.stamp()
            if self.sig is not None and sig != self.sig:
                self.graphs = {}
            self.sig, self.stamp = sig, _keys.stamp()
        entry = self.graphs.get(key)
        if entry is None:
            static_in = [t.clone() for t in inputs]
            # warm-up on a side stream (also triggers lazy initialisations such as ActNorm's)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn(*static_in)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = fn(*static_in)
            entry = (g, static_in, static_out, _packed_blobs(self.owner() if self.owner is not None else None))
            self.graphs[key] = entry
        g, static_in, static_out = entry[:3]
        for dst, src in zip(static_in, inputs):
            dst.copy_(src)
        g.replay()
        # fresh tensors per call: a caller that keeps results of several replays (lps.append(model.log_prob(x))) must not
        # see them all alias the graph's static output
        if torch.is_tensor(static_out):
            return static_out.clone()
        return tuple(t.clone() if torch.is_tensor(t) else t for t in static_out)


class NormalizingFlow(nn.Module):
    """Normalizing flow model (core.py:9-213)."""

    def __init__(self, q0, flows, p=None):
        super().__init__()
        self.q0 = q0
        self.flows = nn.ModuleList(flows)
        self.p = p
        self._graphs = _GraphCache(self)

    # -- MI355X extensions ---------------------------------------------------------------------------------
    def use_graphs(self, mode=True):
        """Replay log_prob / sample_from_noise as hipGraphs (one per batch shape).

        A recorded graph holds the device pointers of the packed weights it was captured with: it is an inference
        feature for frozen parameters.  Graphs are dropped automatically by .to()/.double(), load_state_dict() and
        train(); after changing parameters in place (an optimizer step) call refresh_graphs()."""
        self._graphs.enabled = bool(mode)
        self._graphs.clear()
        return self

    def refresh_graphs(self):
        """Forget the recorded graphs (they are re-captured, with freshly packed weights, on the next call)."""
        self._graphs.clear()
        return self

    def _apply(self, fn, *a, **k):  # .to() / .double() invalidate recorded graphs
        self._graphs.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        self._graphs.clear()
        return super().train(mode)

    # -- reference API -------------------------------------------------------------------------------------
    def forward(self, z):
        for flow in self.flows:
            z, _ = flow(z)
        return z

    def forward_and_log_det(self, z):
        log_det = torch.zeros(len(z), dtype=z.dtype, device=z.device)
        z = run_chain(self.flows, z, False, log_det, +1)
        return z, log_det

    def inverse(self, x):
        for i in range(len(self.flows) - 1, -1, -1):
            x, _ = self.flows[i].inverse(x)
        return x

    def inverse_and_log_det(self, x):
        log_det = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        x = run_chain(self.flows, x, True, log_det, +1)
        return x, log_det

    def _train_pad_rows(self, x):
        """Rows of zero padding that put a differentiable density pass on the 64-row-tile training kernels (config.train_pad_batch):
        > 0 only for a ragged batch of >= 1024 rows through a model with a benchmark-shaped [CoupledRQS, LULinearPermute] pair."""
        B = x.shape[0] if x.dim() == 2 else 0
        if (B < 1024 or B % 64 == 0 or not _config.train_pad_batch or not x.is_cuda or x.dtype != torch.float32 or x.shape[1] != 64
                or not torch.is_grad_enabled() or not (_config.train_pair and _config.train_bwd_onecall)):
            return 0
        fl = list(self.flows)
        for a, b in zip(fl[:-1], fl[1:]):
            if (isinstance(a, CoupledRationalQuadraticSpline) and isinstance(b, LULinearPermute) and b.linear.features == 64
                    and a.prqct._train_full_ok(x, None, False) and all(p.requires_grad for p in a.parameters())
                    and all(p.requires_grad for p in b.parameters())):
                return (-B) % 64
        return 0

    def _log_prob_impl(self, x):
        pad = self._train_pad_rows(x)
        if pad:       # whole 64-row tiles for the training kernels; the slice's backward gives the padding rows a zero cotangent
            B = x.shape[0]
            x = torch.cat([x, x.new_zeros((pad, x.shape[1]))])
        log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        z = run_chain(self.flows, x, True, log_q, +1)
        if hasattr(self.q0, "_log_prob_acc"):
            self.q0._log_prob_acc(z, log_q, +1)
        else:
            log_q += self.q0.log_prob(z)
        return log_q[:B] if pad else log_q

    def log_prob(self, x):
        """log q(x): every layer's inverse, accumulated log-dets, base log-density (core.py:182-197)."""
        if self._graphs.enabled and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._log_prob_impl(x)  # training: autograd path, never recorded into a graph
        return self._graphs.run(("log_prob", tuple(x.shape), x.dtype), self._log_prob_impl, x)

    def forward_kld(self, x):
        """-mean(log q(x)) (core.py:87-102); differentiable when the parameters require gradients."""
        return -torch.mean(self.log_prob(x))

    def _sample_impl(self, eps):
        z, log_q = self.q0.from_noise(eps)
        z = run_chain(self.flows, z, False, log_q, -1)
        return z, log_q

    def sample_from_noise(self, eps):
        """sample() with the base standard-normal noise given (deterministic; used by parity tests)."""
        if self._graphs.enabled and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._sample_impl(eps)
        return self._graphs.run(("sample", tuple(eps.shape), eps.dtype), self._sample_impl, eps)

    def sample(self, num_samples=1):
        """(x, log q(x)) for x ~ q (core.py:167-180)."""
        if hasattr(self.q0, "from_noise"):
            eps = torch.randn((num_samples,) + tuple(self.q0.shape), dtype=self.q0.loc.dtype,
                              device=self.q0.loc.device)
            return self.sample_from_noise(eps)
        z, log_q = self.q0(num_samples)
        for flow in self.flows:
            z = run_flow(flow, z, False, log_q, -1)
        return z, log_q

    def reverse_kld(self, num_samples=1, beta=1.0, score_fn=True, eps=None):
        """mean(log q) - beta mean(log p) on samples of q (core.py:104-131); differentiable through the sampling path
        (reparametrised base sample + the layers' autograd Functions).  score_fn=False re-evaluates log q with the
        parameters frozen (arXiv 1703.09194), as the reference does.  `eps` fixes the base noise (tests)."""
        if eps is None:
            z, log_q = self.sample(num_samples)
        else:
            z, log_q = self.sample_from_noise(eps)
        if not score_fn:
            with _frozen_parameters(self):
                log_q = torch.zeros(len(z), dtype=z.dtype, device=z.device)
                z_ = run_chain(self.flows, z, True, log_q, +1)
                log_q = log_q + self.q0.log_prob(z_)
        log_p = self.p.log_prob(z)
        return torch.mean(log_q) - beta * torch.mean(log_p)

    def reverse_alpha_div(self, num_samples=1, alpha=1, dreg=False, eps=None):
        """Alpha divergence estimated on samples of q (core.py:133-165).  Plain estimator: sign(alpha - 1) logsumexp(alpha log w)
        with log w = log p - log q along the (differentiable) sampling path.  dreg=True: the doubly reparametrised estimator
        (arXiv 1810.04152) -- the importance weights enter as constants, log w is re-evaluated through the density direction with
        the parameters frozen.  `eps` fixes the base noise (tests)."""
        if eps is None:
            z, log_q = self.sample(num_samples)
        else:
            z, log_q = self.sample_from_noise(eps)
        log_p = self.p.log_prob(z)
        if not dreg:
            return (1.0 if alpha > 1 else (-1.0 if alpha < 1 else 0.0)) * torch.logsumexp(alpha * (log_p - log_q), 0)
        w_const = torch.exp(log_p - log_q).detach()
        with _frozen_parameters(self):
            log_q = torch.zeros(len(z), dtype=z.dtype, device=z.device)
            z_ = run_chain(self.flows, z, True, log_q, +1)
            log_q = log_q + self.q0.log_prob(z_)
        w_alpha = w_const ** alpha
        w_alpha = w_alpha / torch.mean(w_alpha)
        weights = (1 - alpha) * w_alpha + alpha * w_alpha ** 2
        return -alpha * torch.mean(weights * (log_p - log_q))

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path))


class ConditionalNormalizingFlow(NormalizingFlow):
    """Normalizing flow whose layers and base distribution receive a context (core.py:216-366): same loops, every
    call carries `context=context`; log-dets are folded into the accumulator by the kernels (Flow._run)."""

    def forward(self, z, context=None):
        for flow in self.flows:
            z, _ = flow(z, context=context)
        return z

    def forward_and_log_det(self, z, context=None):
        log_det = torch.zeros(len(z), dtype=z.dtype, device=z.device)
        for flow in self.flows:
            z = run_flow(flow, z, False, log_det, +1, context=context)
        return z, log_det

    def inverse(self, x, context=None):
        for i in range(len(self.flows) - 1, -1, -1):
            x, _ = self.flows[i].inverse(x, context=context)
        return x

    def inverse_and_log_det(self, x, context=None):
        log_det = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        for i in range(len(self.flows) - 1, -1, -1):
            x = run_flow(self.flows[i], x, True, log_det, +1, context=context)
        return x, log_det

    def sample(self, num_samples=1, context=None):
        z, log_q = self.q0(num_samples, context=context)
        for flow in self.flows:
            z = run_flow(flow, z, False, log_q, -1, context=context)
        return z, log_q

    def log_prob(self, x, context=None):
        log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        z = x
        for i in range(len(self.flows) - 1, -1, -1):
            z = run_flow(self.flows[i], z, True, log_q, +1, context=context)
        log_q += self.q0.log_prob(z, context=context)
        return log_q

    def forward_kld(self, x, context=None):
        return -torch.mean(self.log_prob(x, context=context))

    def reverse_kld(self, num_samples=1, context=None, beta=1.0, score_fn=True):
        """core.py:338-366; score_fn=False re-evaluates log q with the parameters frozen, as the reference does."""
        z, log_q = self.sample(num_samples, context=context)
        if not score_fn:
            with _frozen_parameters(self):
                log_q = torch.zeros(len(z), dtype=z.dtype, device=z.device)
                z_ = z
                for i in range(len(self.flows) - 1, -1, -1):
                    z_ = run_flow(self.flows[i], z_, True, log_q, +1, context=context)
                log_q = log_q + self.q0.log_prob(z_, context=context)
        log_p = self.p.log_prob(z, context=context)
        return torch.mean(log_q) - beta * torch.mean(log_p)


class MultiscaleFlow(nn.Module):
    """Multi-scale (RealNVP / Glow) flow (core.py:455-655)."""

    def __init__(self, q0, flows, merges, transform=None, class_cond=True):
        super().__init__()
        self.q0 = nn.ModuleList(q0)
        self.num_levels = len(self.q0)
        self.flows = torch.nn.ModuleList([nn.ModuleList(flow) for flow in flows])
        self.merges = torch.nn.ModuleList(merges)
        self.transform = transform
        self.class_cond = class_cond
        self._graphs = _GraphCache(self)

    def use_graphs(self, mode=True):
        """Replay log_prob (without class labels) as a hipGraph per input shape: a Glow pass is ~100 small launches, i.e.
        launch-bound when issued eagerly.

        A recorded graph holds the device pointers of the packed weights it was captured with (ConvNet2d / GlowBlock /
        Invertible1x1Conv blobs, keyed by parameter version): an inference feature for FROZEN parameters.  Graphs are
        dropped by use_graphs(), .to()/.double(), load_state_dict() and train(); after changing parameters in place (an
        optimizer step) call refresh_graphs().  The blobs a graph was captured with stay referenced by the graph entry,
        so a stale replay can never read freed memory."""
        self._graphs.enabled = bool(mode)
        self._graphs.clear()
        return self

    def refresh_graphs(self):
        self._graphs.clear()
        return self

    def _apply(self, fn, *a, **k):
        self._graphs.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        self._graphs.clear()
        return super().train(mode)

    def forward_kld(self, x, y=None):
        return -torch.mean(self.log_prob(x, y))

    def forward(self, x, y=None):
        return -self.log_prob(x, y)

    def forward_and_log_det(self, z):
        log_det = torch.zeros(len(z[0]), dtype=z[0].dtype, device=z[0].device)
        for i in range(len(self.q0)):
            if i == 0:
                z_ = z[0]
            else:
                z_, _ = self.merges[i - 1]([z_, z[i]])
            for flow in self.flows[i]:
                z_ = run_flow(flow, z_, False, log_det, +1)
        if self.transform is not None:
            z_ = run_flow(self.transform, z_, False, log_det, +1)
        return z_, log_det

    def inverse_and_log_det(self, x):
        log_det = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        if self.transform is not None:
            x = run_flow(self.transform, x, True, log_det, +1)
        z = [None] * len(self.q0)
        for i in range(len(self.q0) - 1, -1, -1):
            for flow in reversed(self.flows[i]):
                x = run_flow(flow, x, True, log_det, +1)
            if i == 0:
                z[i] = x
            else:
                [x, z[i]], _ = self.merges[i - 1].inverse(x)
        return z, log_det

    def _level_pass(self, i, z, z_other, inverse, ld, acc):
        """_level_pass_impl; in the density direction under autograd the level's Invertible1x1Convs assemble their matrices in one
        launch first (flows/mixing.prefetch_weights, round 6)."""
        convs, cnets = [], []
        if inverse and torch.is_grad_enabled() and z.is_cuda:
            from . import nets as _nets
            from .flows.glow import GlowBlock
            from .flows.mixing import Invertible1x1Conv, clear_prefetched, prefetch_weights
            blocks = [f for f in self.flows[i] if isinstance(f, GlowBlock) and len(f.flows) == 3]
            convs = [f.flows[1] for f in blocks if isinstance(f.flows[1], Invertible1x1Conv)]
            if len(convs) > 1:
                prefetch_weights(convs)
            # ... and the conditioners' packed weight streams of this step (nets.prefetch_train_packs: one gather launch per structure)
            cnets = [getattr(f.flows[0].flows[1], "param_map", None) for f in blocks
                     if hasattr(f.flows[0], "flows") and len(f.flows[0].flows) == 3]
            cnets = [c for c in cnets if isinstance(c, _nets.ConvNet2d) and any(p.requires_grad for p in c.parameters())]
            if len(cnets) > 1:
                _nets.prefetch_train_packs(cnets, z.device)
        from .flows.affine import lazy_ld
        try:
            with lazy_ld(ld):          # (the level's layers only ADD to ld: their statements go out as one launch, same order)
                return self._level_pass_impl(i, z, z_other, inverse, ld, acc)
        finally:
            if len(convs) > 1:
                clear_prefetched(convs)
            if len(cnets) > 1:
                _nets.clear_prefetched_packs(cnets)

    def _level_pass_impl(self, i, z, z_other, inverse, ld, acc):
        """Level i of the multi-scale flow with the log-dets folded into `ld`.
        inverse (density, core.py:600-611): flows[i] reversed, then the level's split -> (z, z_); z_ is None at i == 0.
        forward (sample, core.py:570-582): merge z with `z_other` (None at i == 0), then flows[i] -> (z, None).
        Runs of GlowBlocks of one shape go out as ONE persistent launch (nf_glow_level) with the neighbouring
        Squeeze / channel Split / Merge folded into its load and store; anything else goes layer by layer."""
        from .autograd import needs_grad
        from .flows.glow import GlowBlock, plan_level, run_level
        from .flows.reshape import Merge, Squeeze
        flows = list(self.flows[i])
        seq = flows[::-1] if inverse else flows
        merge = self.merges[i - 1] if i > 0 else None
        chan = merge is not None and type(merge) is Merge and merge.mode == "channel"
        from . import _prof, config
        fast = (config.glow_level_chains and z.is_cuda and z.dtype == torch.float32 and z.dim() == 4
                and not needs_grad(z, z_other) and (z_other is None or z_other.dtype == torch.float32))
        pending_merge = (not inverse) and merge is not None      # sample direction: the merge is not applied yet
        refused = self.__dict__.setdefault("_level_refused", set())
        k = 0
        while k < len(seq):
            f = seq[k]
            if fast:
                sq_in = (inverse and isinstance(f, Squeeze) and k + 1 < len(seq) and isinstance(seq[k + 1], GlowBlock)
                         and z.shape[2] % 2 == 0 and z.shape[3] % 2 == 0)
                start = k + 1 if sq_in else k
                if isinstance(seq[start], GlowBlock) and not (pending_merge and not chan):
                    B = z.shape[0]
                    if pending_merge:
                        C, H, W = z.shape[1] + z_other.shape[1], z.shape[2], z.shape[3]
                    elif sq_in:
                        C, H, W = 4 * z.shape[1], z.shape[2] // 2, z.shape[3] // 2
                    else:
                        C, H, W = z.shape[1:]
                    key = (i, inverse, start, B, C, H, W)
                    n = 0
                    if key not in refused:
                        n, entries, layout, slope, smap = plan_level(seq[start:], B, C, H, W, inverse)
                    if n:
                        end = start + n
                        sq_out = (not inverse) and end < len(seq) and isinstance(seq[end], Squeeze) and C % 4 == 0
                        split_out = inverse and end == len(seq) and chan
                        try:
                            with _prof.range_("glow_level[%d x GlowBlock %dx%dx%d].%s" % (
                                    n, C, H, W, "inverse" if inverse else "forward")):
                                out0, out1 = run_level(seq[start:end], entries, layout, slope, smap, z,
                                                       z_other if pending_merge else None, sq_in, inverse, ld, acc,
                                                       cout0=(C + 1) // 2 if split_out else None, out_squeezed=sq_out)
                        except NotImplementedError:     # working set beyond one workgroup's LDS: nothing was launched
                            refused.add(key)
                        else:
                            pending_merge = False
                            k = end + (1 if sq_out else 0)
                            z = out0
                            if split_out:
                                return out0, out1
                            continue
            if pending_merge:
                z, _ = merge([z, z_other])
                pending_merge = False
            z = run_flow(f, z, inverse, ld, acc)
            k += 1
        if pending_merge:
            z, _ = merge([z, z_other])
        if inverse and merge is not None:
            [z, z_], _ = merge.inverse(z)
            return z, z_
        return z, None

    def sample(self, num_samples=1, y=None, temperature=None):
        if temperature is not None:
            self.set_temperature(temperature)
        for i in range(len(self.q0)):
            if self.class_cond:
                z_, log_q_ = self.q0[i](num_samples, y)
            else:
                z_, log_q_ = self.q0[i](num_samples)
            if i == 0:
                log_q = log_q_
                z, _ = self._level_pass(0, z_, None, False, log_q, -1)
            else:
                log_q += log_q_
                z, _ = self._level_pass(i, z, z_, False, log_q, -1)
        if self.transform is not None:
            z = run_flow(self.transform, z, False, log_q, -1)
        if temperature is not None:
            self.reset_temperature()
        return z, log_q

    def sample_from_noise(self, eps):
        """sample() (core.py:553-586) with the base noise of every level given: `eps[i]` has the shape of q0[i]'s sample
        (deterministic; used by the parity tests against reference fixtures).  DiagGaussian bases only."""
        for i in range(len(self.q0)):
            z_, log_q_ = self.q0[i].from_noise(eps[i])
            if i == 0:
                log_q = log_q_
                z, _ = self._level_pass(0, z_, None, False, log_q, -1)
            else:
                log_q = log_q + log_q_
                z, _ = self._level_pass(i, z, z_, False, log_q, -1)
        if self.transform is not None:
            z = run_flow(self.transform, z, False, log_q, -1)
        return z, log_q

    def log_prob(self, x, y=None):
        """core.py:588-616."""
        if y is None and self._graphs.enabled and not torch.is_grad_enabled():
            return self._graphs.run(("log_prob", tuple(x.shape), x.dtype), self._log_prob_impl, x)
        return self._log_prob_impl(x, y)

    def _log_prob_impl(self, x, y=None):
        log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        z = x
        if self.transform is not None:
            z = run_flow(self.transform, z, True, log_q, +1)
        for i in range(len(self.q0) - 1, -1, -1):
            z, z_ = self._level_pass(i, z, None, True, log_q, +1)
            if i == 0:
                z_ = z
            if self.class_cond:
                log_q += self.q0[i].log_prob(z_, y)
            elif hasattr(self.q0[i], "_log_prob_acc"):
                self.q0[i]._log_prob_acc(z_.contiguous(), log_q, +1)
            else:
                log_q += self.q0[i].log_prob(z_)
        return log_q

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path))

    def set_temperature(self, temperature):
        for q0 in self.q0:
            if hasattr(q0, "temperature"):
                q0.temperature = temperature
            else:
                raise NotImplementedError("One base function does not support temperature annealed sampling")

    def reset_temperature(self):
        self.set_temperature(None)
