"""Conditioner networks (L1 of the reference): ResidualNet / ResidualBlock, MLP, ConvNet2d.

Mirrors normflows/nets/resnet.py:7-104, nets/mlp.py:5-58, nets/cnn.py:5-63: constructor signatures, parameter
registration and RNG consumption order (a seeded build reproduces the reference's initial weights) and
state_dict keys.  These modules are the MFMA-shaped part of the path.  `forward` here issues plain library
GEMMs / convolutions through torch (rocBLAS / hipBLASLt / MIOpen); the NSF coupling layer bypasses
ResidualNet.forward and feeds the same weights to the hand-written fused MFMA kernel when the shape allows.
"""
import torch
from . import _keys
from torch import nn
from torch.nn import functional as F, init


class Linear(nn.Linear):
    """nn.Linear whose weight / bias gradients use the split-K HIP kernel at training batch sizes (autograd.linear);
    same parameters and state_dict keys."""

    def forward(self, x):
        from . import autograd
        return autograd.linear(x, self.weight, self.bias)


class ResidualBlock(nn.Module):
    """Pre-activation residual block x + W2 act(W1 act(x)) with optional GLU context gate (resnet.py:7-50)."""

    def __init__(self, features, context_features, activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 zero_initialization=True):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList([Linear(features, features) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
            init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        if context is None:
            from . import autograd
            if autograd.residual_block_fused_ok(self, inputs):   # training: one launch forward, one backward (nf_rows_block)
                l1, l2 = self.linear_layers
                return autograd.ResidualBlockFn.apply(inputs, l1.weight, l1.bias, l2.weight, l2.bias)
        temps = inputs
        if self.use_batch_norm:
            temps = self.batch_norm_layers[0](temps)
        temps = self.linear_layers[0](self.activation(temps))
        if self.use_batch_norm:
            temps = self.batch_norm_layers[1](temps)
        temps = self.linear_layers[1](self.dropout(self.activation(temps)))
        if context is not None:
            temps = F.glu(torch.cat((temps, self.context_layer(context)), dim=1), dim=1)
        return inputs + temps


def _train_packs_from(module, build, params, device, skey=()):
    """(forward pack, backward pack) for autograd.MadeFn / ConvNetFn: the value-independent structure (tables, gather indices:
    flows/made_pack.train_structure) is built once per module and device; the weight streams are gathered on the device from the
    parameters as they are in THIS call (nf_pack_gather) -- under autograd they change every step."""
    struct = _train_struct_for(module, build, device, skey)
    if struct is None:
        return None
    from . import ops
    bwd = dict(struct["bwd"])
    both = module.__dict__.pop("_both_prefetch", None)        # (gathered with the level's other conditioners: prefetch_train_packs)
    if both is None or both.numel() != struct["src"].numel():
        both = ops.pack_gather(params, struct["src"])
    bwd["blob"] = both[struct["nfwd"]:]
    return (both[:struct["nfwd"]], struct["table"], struct["hp"]), bwd


def prefetch_train_packs(convnets, device):
    """Round 6 (last session): the packed weight streams of several ConvNet2d conditioners of ONE structure for this training step in
    one launch (ops.pack_gather_batch) -- they depend on parameters only, so a Glow level gathers its K conditioners' streams before its
    first block runs.  Each module's result waits in `_both_prefetch` for its next `_train_packs`; the caller drops what was not
    consumed (clear_prefetched_packs)."""
    from . import config, ops
    if not config.glow_weights_batched or not torch.is_grad_enabled():
        return
    groups = {}
    for net in convnets:
        args = net._train_pack_args() if hasattr(net, "_train_pack_args") else None
        if args is None:
            continue
        build, params, key = args
        if not all(p.is_cuda and p.device == device and p.dtype == torch.float32 and p.is_contiguous() for p in params):
            continue
        struct = _train_struct_for(net, build, device)
        if struct is None:
            continue
        groups.setdefault(key, []).append((net, params, struct))
    for items in groups.values():
        if len(items) < 2:
            continue
        src = items[0][2]["src"]
        if any(it[2]["src"].numel() != src.numel() for it in items):
            continue
        outs = ops.pack_gather_batch([[p.detach() for p in it[1]] for it in items], src)
        for it, o in zip(items, outs):
            it[0].__dict__["_both_prefetch"] = o


def clear_prefetched_packs(convnets):
    for net in convnets:
        net.__dict__.pop("_both_prefetch", None)


def _train_struct_for(module, build, device, skey=()):
    """The module's value-independent pack structure with device copies of its tables and gather indices (built once)."""
    st = module.__dict__.get("_train_struct")
    skey = (str(device),) + tuple(skey)             # (skey: what the structure itself depends on, e.g. MADE's mask buffers)
    if st is None or st[0] != skey:
        struct = build()
        if struct is not None:
            bwd = struct["bwd"]
            for k in ("table", "wtable", "stable", "mask", "src"):
                bwd[k] = torch.from_numpy(bwd[k]).to(device)
            bwd.pop("blob", None)
            struct["table"] = torch.from_numpy(struct["table"]).to(device)
            struct["nfwd"] = int(struct["src"].size)             # one gather for both streams: [forward | backward]
            struct["src"] = torch.cat([torch.from_numpy(struct["src"]), bwd["src"].cpu()]).to(device)
        st = module.__dict__["_train_struct"] = (skey, struct)
    return st[1]


class ResidualNet(nn.Module):
    """initial Linear -> num_blocks ResidualBlocks -> final Linear (resnet.py:53-104)."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None, num_blocks=2,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False, preprocessing=None):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        self.preprocessing = preprocessing
        if context_features is not None:
            self.initial_layer = Linear(in_features + context_features, hidden_features)
        else:
            self.initial_layer = Linear(in_features, hidden_features)
        self.blocks = nn.ModuleList([
            ResidualBlock(features=hidden_features, context_features=context_features, activation=activation,
                          dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)
        ])
        self.final_layer = Linear(hidden_features, out_features)
        self.dropout_probability = dropout_probability
        self.use_batch_norm = use_batch_norm

    def _train_packs(self, device):
        """Packs for the MADE training kernels (dense); None outside their structure.  Eligibility is re-checked on EVERY call:
        `is_plain_relu` depends on the mode when dropout_probability > 0 (eval: dropout is the identity, train: it is not), so
        neither a structure built in eval() may serve a later train() step (MadeFn has no dropout: it would be skipped
        silently) nor may a `None` seen in train() stick to the module (nets.py:78)."""
        from .flows import made_pack
        if not made_pack.resnet_supported(self):
            return None
        lins = [self.initial_layer] + [l for b in self.blocks for l in b.linear_layers] + [self.final_layer]
        return _train_packs_from(self, lambda: made_pack.resnet_train_structure(self), [t for l in lins for t in (l.weight, l.bias)],
                                 device)

    def forward(self, inputs, context=None):
        # hidden widths beyond the 128-column training kernels (ResidualBlockFn, autograd.linear) under autograd: the whole net's
        # forward, input-gradient chain and weight gradients on the MADE training kernels (csrc/made_bwd.hip; no mask = dense)
        if (context is None and self.hidden_features > 128 and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))):
            from . import config
            packs = self._train_packs(inputs.device) if (config.made_train and config.made_fused) else None
            if packs is not None:
                from . import autograd
                lins = [self.initial_layer] + [l for b in self.blocks for l in b.linear_layers] + [self.final_layer]
                return autograd.MadeFn.apply(packs[0], packs[1], inputs, *[t for l in lins for t in (l.weight, l.bias)])
        temps = inputs if self.preprocessing is None else self.preprocessing(inputs)
        if context is None:
            temps = self.initial_layer(temps)
        else:
            temps = self.initial_layer(torch.cat((temps, context), dim=1))
        for block in self.blocks:
            temps = block(temps, context=context)
        return self.final_layer(temps)

    def is_plain_relu(self):
        """True when the net is the plain ReLU MLP the fused HIP kernel implements."""
        act_ok = all(isinstance(b.activation, nn.ReLU) or b.activation is F.relu for b in self.blocks)
        return (act_ok and self.context_features is None and self.preprocessing is None and not self.use_batch_norm
                and (self.dropout_probability == 0.0 or not self.training))


class PeriodicFeaturesElementwise(nn.Module):
    """Replaces the features `ind` by w1 sin(scale f) + w2 cos(scale f) (utils/nn.py:64-129): preprocessing of the
    conditioner for circular coordinates (plain tensor arithmetic in front of the GEMMs)."""

    def __init__(self, ndim, ind, scale=1.0, bias=False, activation=None):
        super().__init__()
        self.ndim = ndim
        if torch.is_tensor(ind):
            self.register_buffer("ind", ind.long())
        else:
            self.register_buffer("ind", torch.tensor(ind, dtype=torch.long))
        ind_ = [i for i in range(self.ndim) if i not in self.ind]
        self.register_buffer("ind_", torch.tensor(ind_, dtype=torch.long))
        perm_ = torch.cat((self.ind, self.ind_))
        inv_perm_ = torch.zeros_like(perm_)
        for i in range(self.ndim):
            inv_perm_[perm_[i]] = i
        self.register_buffer("inv_perm", inv_perm_)
        self.weights = nn.Parameter(torch.ones(len(self.ind), 2))
        if torch.is_tensor(scale):
            self.register_buffer("scale", scale)
        else:
            self.scale = scale
        self.apply_bias = bias
        if self.apply_bias:
            self.bias = nn.Parameter(torch.zeros(len(self.ind)))
        self.activation = torch.nn.Identity() if activation is None else activation

    def forward(self, inputs):
        inputs_ = self.scale * inputs[..., self.ind]
        inputs_ = self.weights[:, 0] * torch.sin(inputs_) + self.weights[:, 1] * torch.cos(inputs_)
        if self.apply_bias:
            inputs_ = inputs_ + self.bias
        inputs_ = self.activation(inputs_)
        out = torch.cat((inputs_, inputs[..., self.ind_]), -1)
        return out[..., self.inv_perm]


class ConstScaleLayer(nn.Module):
    """utils/nn.py ConstScaleLayer: multiply by a constant."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale_cpu = torch.tensor(scale)
        self.register_buffer("scale", self.scale_cpu)

    def forward(self, input):
        return input * self.scale


class ClampExp(nn.Module):
    """utils/nn.py ClampExp: exp(min(x, 1))."""

    def forward(self, x):
        one = torch.tensor(1.0, device=x.device, dtype=x.dtype)
        return torch.exp(torch.min(x, one))


class MLP(nn.Module):
    """Linear + LeakyReLU stack (mlp.py:5-58)."""

    def __init__(self, layers, leaky=0.0, score_scale=None, output_fn=None, output_scale=None, init_zeros=False,
                 dropout=None):
        super().__init__()
        net = nn.ModuleList([])
        for k in range(len(layers) - 2):
            net.append(nn.Linear(layers[k], layers[k + 1]))
            net.append(nn.LeakyReLU(leaky))
        if dropout is not None:
            net.append(nn.Dropout(p=dropout))
        net.append(nn.Linear(layers[-2], layers[-1]))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        if output_fn is not None:
            if score_scale is not None:
                net.append(ConstScaleLayer(score_scale))
            if output_fn == "sigmoid":
                net.append(nn.Sigmoid())
            elif output_fn == "relu":
                net.append(nn.ReLU())
            elif output_fn == "tanh":
                net.append(nn.Tanh())
            elif output_fn == "clampexp":
                net.append(ClampExp())
            else:
                raise NotImplementedError("This output function is not implemented.")
            if output_scale is not None:
                net.append(ConstScaleLayer(output_scale))
        self.net = nn.Sequential(*net)

    def forward(self, x):
        return self.net(x)


class _NetActNorm(nn.Module):
    """utils/nn.py ActNorm wrapper: the flow's forward output without the log-det."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        from .flows.normalization import ActNorm
        self.actNorm = ActNorm(*args, **kwargs)

    def forward(self, input):
        out, _ = self.actNorm(input)
        return out


class ConvNet2d(nn.Module):
    """Conv2d + LeakyReLU stack used by GlowBlock (cnn.py:5-63)."""

    def __init__(self, channels, kernel_size, leaky=0.0, init_zeros=True, actnorm=False, weight_std=None):
        super().__init__()
        net = nn.ModuleList([])
        for i in range(len(kernel_size) - 1):
            conv = nn.Conv2d(channels[i], channels[i + 1], kernel_size[i], padding=kernel_size[i] // 2,
                             bias=(not actnorm))
            if weight_std is not None:
                conv.weight.data.normal_(mean=0.0, std=weight_std)
            net.append(conv)
            if actnorm:
                net.append(_NetActNorm((channels[i + 1],) + (1, 1)))
            net.append(nn.LeakyReLU(leaky))
        i = len(kernel_size)
        net.append(nn.Conv2d(channels[i - 1], channels[i], kernel_size[i - 1], padding=kernel_size[i - 1] // 2))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        self.net = nn.Sequential(*net)

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled():
            fused = self._fused_pack(x)
            if fused is not None:
                from . import ops
                return ops.glow_convnet(x, fused[0], self.net[-1].out_channels, self.net[1].negative_slope, fused[1])
            return self._forward_inference(x)
        from . import autograd
        # (only when a gradient is actually wanted -- a grad-enabled call on frozen weights and a plain input takes the
        # library path instead of a Function whose backward would never run)
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and autograd.needs_grad(x, *self.parameters()):
            packs = self._train_packs(x.device)
            if packs is not None:
                from . import autograd
                c1, c2, c3 = self.net[0], self.net[2], self.net[4]
                return autograd.ConvNetFn.apply(packs[0], packs[1], x, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias)
        return self.net(x)

    def _train_packs(self, device):
        """Device copies of the plain-MLP packs (flows/made_pack.pack_mlp_*) of the 3x3 -> 1x1 -> 3x3 ReLU network for the MADE training
        kernels (autograd.ConvNetFn), rebuilt when a parameter changes; None for any other structure (then the library path)."""
        args = self._train_pack_args()
        if args is None:
            return None
        return _train_packs_from(self, args[0], args[1], device)

    def _train_pack_args(self):
        """(structure builder, parameter list, structure key) of _train_packs, or None."""
        from . import config
        mods = list(self.net)
        if not (config.made_train and config.made_fused) or len(mods) != 5:
            return None
        c1, a1, c2, a2, c3 = mods
        if not (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d)
                and isinstance(a1, nn.LeakyReLU) and isinstance(a2, nn.LeakyReLU)):
            return None
        if (c1.kernel_size, c2.kernel_size, c3.kernel_size) != ((3, 3), (1, 1), (3, 3)) or any(c.bias is None for c in (c1, c2, c3)):
            return None
        if a1.negative_slope != 0.0 or a2.negative_slope != 0.0 or c1.weight.dtype != torch.float32:
            return None
        if any(c.padding != (k // 2, k // 2) or c.stride != (1, 1) or c.dilation != (1, 1) or c.groups != 1
               or c.padding_mode != "zeros"          # (the gather kernels pad with zeros)
               for c, k in ((c1, 3), (c2, 1), (c3, 3))):
            return None
        hid, cin, cout = c1.out_channels, c1.in_channels, c3.out_channels
        if not (c2.in_channels == hid and c2.out_channels == hid and c3.in_channels == hid):
            return None
        from .flows import made_pack
        return (lambda: made_pack.convnet_train_structure(cin, hid, cout), [c1.weight, c1.bias, c2.weight, c2.bias, c3.weight],
                ("conv", cin, hid, cout))

    # below this many pixels per call the library path stays (the one-launch kernels are built for full-chip batches)
    FUSED_MIN_PIXELS = 2048
    FUSED_WIDE_MIN_PIXELS = 64 * 256    # the 256-pixel-workgroup kernel's run time does not depend on the batch (one
                                        # round of workgroups): it overtakes the library path at about 64 workgroups

    def _fused_pack(self, x):
        """(packed weights, layout) for ops.glow_convnet when this is the GlowBlock network (3x3 -> 1x1 -> 3x3 around 256
        hidden channels, biases, equal LeakyReLU slopes) and the call is one a kernel takes; None otherwise.  Repacked
        when a parameter changes; one pack per layout (256-, 64- or 16-pixel workgroups)."""
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
            return None
        B, Cin, H, W = x.shape
        if x.stride(3) != 1 or x.stride(2) != W or x.stride(1) != H * W:
            return None
        return self._fused_pack_for(B, H, W)

    def _fused_pack_for(self, B, H, W):
        """_fused_pack for a float32 device input of shape (B, Cin, H, W) that need not exist yet (level chains)."""
        mods = list(self.net)
        if len(mods) != 5 or torch.is_grad_enabled():
            return None
        c1, a1, c2, a2, c3 = mods
        if not (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d)
                and isinstance(a1, nn.LeakyReLU) and isinstance(a2, nn.LeakyReLU)):
            return None
        if (c1.kernel_size, c2.kernel_size, c3.kernel_size) != ((3, 3), (1, 1), (3, 3)):
            return None
        hid = c1.out_channels
        if hid > 256 or c2.out_channels != hid or c2.in_channels != hid or c3.in_channels != hid:
            return None      # the kernels' hidden width is 256: narrower networks run on them zero-padded (below)
        if any(c.bias is None for c in (c1, c2, c3)):
            return None
        if a1.negative_slope != a2.negative_slope or not 0.0 <= a1.negative_slope <= 1.0:
            return None
        if not c1.weight.is_cuda or c1.weight.dtype != torch.float32:
            return None
        from . import ops
        if B * H * W < self.FUSED_MIN_PIXELS:
            return None
        layout = ops.glow_convnet_layout(B, H, W)
        if layout is None or (layout == ops.GLOW_CONV_WIDE and B * H * W < self.FUSED_WIDE_MIN_PIXELS):
            return None
        params = [c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias]
        key = _keys.pkey(params)
        cache = self.__dict__.setdefault("_gc_cache", {})
        hit = cache.get(layout)
        if hit is None or hit[0] != key:
            ws = [p_.detach() for p_ in params]
            if hid != 256:   # zero hidden channels stay zero through the bias-free LeakyReLU: pad up to the kernels' 256
                w1, b1, w2, b2, w3, b3 = ws
                W1 = w1.new_zeros(256, w1.shape[1], 3, 3)
                W1[:hid] = w1
                B1 = b1.new_zeros(256)
                B1[:hid] = b1
                W2 = w2.new_zeros(256, 256, 1, 1)
                W2[:hid, :hid] = w2
                B2 = b2.new_zeros(256)
                B2[:hid] = b2
                W3 = w3.new_zeros(w3.shape[0], 256, 3, 3)
                W3[:, :hid] = w3
                ws = [W1, B1, W2, B2, W3, b3]
            hit = cache[layout] = (key, ops.glow_convnet_pack(*ws, layout=layout))
        return None if hit[1] is None else (hit[1], layout)

    def forward_split(self, x):
        """(output without the last convolution's bias, that bias) for callers that fold the bias into their own kernel
        (the affine coupling); None when not applicable."""
        last = self.net[-1]
        if (torch.is_grad_enabled() or not x.is_cuda or not isinstance(last, nn.Conv2d) or last.bias is None
                or x.dtype not in (torch.float32, torch.float64)):
            return None
        if self._fused_pack(x) is not None:
            return None      # the one-launch kernel adds the bias itself: callers use forward()
        h = self._forward_inference(x, upto=len(self.net) - 1)
        return F.conv2d(h, last.weight, None, last.stride, last.padding, last.dilation, last.groups), last.bias

    def _forward_inference(self, x, upto=None):
        """Library convolution without bias, then ONE HIP pass for bias + LeakyReLU (instead of a bias pass inside the
        library call and a separate activation pass)."""
        from . import ops
        mods = list(self.net) if upto is None else list(self.net)[:upto]
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (isinstance(m, nn.Conv2d) and isinstance(nxt, nn.LeakyReLU) and m.bias is not None
                    and x.dtype in (torch.float32, torch.float64)):
                x = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
                x = ops.bias_leaky_relu_(x.contiguous(), m.bias, nxt.negative_slope)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


# ---- MADE (conditioner of the autoregressive flows) ---------------------------------------------------------------
def _tile(x, n):
    """utils/nn.py tile: repeat every element n times, keeping order."""
    return x.reshape(-1).repeat(n).reshape(n, -1).transpose(1, 0).reshape(-1)


def _get_input_degrees(in_features):
    return torch.arange(1, in_features + 1)


class MaskedLinear(nn.Linear):
    """Linear layer whose weight is multiplied by a fixed autoregressive mask (nets/made.py:19-81).  The masked
    weight is cached per parameter version, so the D sequential passes of the MAF inverse do not re-mask it."""

    def __init__(self, in_degrees, out_features, autoregressive_features, random_mask, is_output, bias=True,
                 out_degrees_=None):
        super().__init__(in_features=len(in_degrees), out_features=out_features, bias=bias)
        mask, degrees = self._get_mask_and_degrees(in_degrees=in_degrees, out_features=out_features,
                                                   autoregressive_features=autoregressive_features,
                                                   random_mask=random_mask, is_output=is_output,
                                                   out_degrees_=out_degrees_)
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", degrees)
        self._masked_cache = None

    @classmethod
    def _get_mask_and_degrees(cls, in_degrees, out_features, autoregressive_features, random_mask, is_output,
                              out_degrees_=None):
        if is_output:
            if out_degrees_ is None:
                out_degrees_ = _get_input_degrees(autoregressive_features)
            out_degrees = _tile(out_degrees_, out_features // autoregressive_features)
            mask = (out_degrees[..., None] > in_degrees).float()
        else:
            if random_mask:
                min_in_degree = torch.min(in_degrees).item()
                min_in_degree = min(min_in_degree, autoregressive_features - 1)
                out_degrees = torch.randint(low=min_in_degree, high=autoregressive_features, size=[out_features],
                                            dtype=torch.long)
            else:
                max_ = max(1, autoregressive_features - 1)
                min_ = min(1, autoregressive_features - 1)
                out_degrees = torch.arange(out_features) % max_ + min_
            mask = (out_degrees[..., None] >= in_degrees).float()
        return mask, out_degrees

    def masked_weight(self):
        if torch.is_grad_enabled() and self.weight.requires_grad:
            return self.weight * self.mask
        key = _keys.pkey((self.weight,)) + (self.mask.data_ptr(),)
        if self._masked_cache is None or self._masked_cache[0] != key:
            self._masked_cache = (key, (self.weight.detach() * self.mask).contiguous())
        return self._masked_cache[1]

    def forward(self, x):
        return F.linear(x, self.masked_weight(), self.bias)


class MaskedFeedforwardBlock(nn.Module):
    """nets/made.py:84-137."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        features = len(in_degrees)
        self.batch_norm = nn.BatchNorm1d(features, eps=1e-3) if use_batch_norm else None
        if context_features is not None:
            raise NotImplementedError()
        self.linear = MaskedLinear(in_degrees=in_degrees, out_features=features,
                                   autoregressive_features=autoregressive_features, random_mask=random_mask,
                                   is_output=False)
        self.degrees = self.linear.degrees
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, inputs, context=None):
        if context is not None:
            raise NotImplementedError()
        outputs = self.batch_norm(inputs) if self.batch_norm else inputs
        return self.dropout(self.activation(self.linear(outputs)))


class MaskedResidualBlock(nn.Module):
    """nets/made.py:140-214."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False, zero_initialization=True):
        if random_mask:
            raise ValueError("Masked residual block can't be used with random masks.")
        super().__init__()
        features = len(in_degrees)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        linear_0 = MaskedLinear(in_degrees=in_degrees, out_features=features,
                                autoregressive_features=autoregressive_features, random_mask=False, is_output=False)
        linear_1 = MaskedLinear(in_degrees=linear_0.degrees, out_features=features,
                                autoregressive_features=autoregressive_features, random_mask=False, is_output=False)
        self.linear_layers = nn.ModuleList([linear_0, linear_1])
        self.degrees = linear_1.degrees
        if torch.all(self.degrees >= in_degrees).item() != 1:
            raise RuntimeError("In a masked residual block, the output degrees can't be less than the corresponding "
                               "input degrees.")
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, a=-1e-3, b=1e-3)
            init.uniform_(self.linear_layers[-1].bias, a=-1e-3, b=1e-3)

    def forward(self, inputs, context=None):
        if context is None:
            from . import autograd
            if autograd.residual_block_fused_ok(self, inputs):   # training: one launch forward, one backward (nf_rows_block)
                l1, l2 = self.linear_layers                      # on the MASKED weights (weight * mask stays in the graph)
                return autograd.ResidualBlockFn.apply(inputs, l1.masked_weight(), l1.bias, l2.masked_weight(), l2.bias)
        temps = inputs
        if self.use_batch_norm:
            temps = self.batch_norm_layers[0](temps)
        temps = self.linear_layers[0](self.activation(temps))
        if self.use_batch_norm:
            temps = self.batch_norm_layers[1](temps)
        temps = self.linear_layers[1](self.dropout(self.activation(temps)))
        if context is not None:
            temps = F.glu(torch.cat((temps, self.context_layer(context)), dim=1), dim=1)
        return inputs + temps


class MADE(nn.Module):
    """Masked autoencoder for distribution estimation (nets/made.py:217-304)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, output_multiplier=1,
                 use_residual_blocks=True, random_mask=False, permute_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False, preprocessing=None):
        if use_residual_blocks and random_mask:
            raise ValueError("Residual blocks can't be used with random masks.")
        super().__init__()
        self.preprocessing = torch.nn.Identity() if preprocessing is None else preprocessing
        input_degrees_ = _get_input_degrees(features)
        if permute_mask:
            input_degrees_ = input_degrees_[torch.randperm(features)]
        self.initial_layer = MaskedLinear(in_degrees=input_degrees_, out_features=hidden_features,
                                          autoregressive_features=features, random_mask=random_mask, is_output=False)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        blocks = []
        block_constructor = MaskedResidualBlock if use_residual_blocks else MaskedFeedforwardBlock
        prev_out_degrees = self.initial_layer.degrees
        for _ in range(num_blocks):
            blocks.append(block_constructor(in_degrees=prev_out_degrees, autoregressive_features=features,
                                            context_features=context_features, random_mask=random_mask,
                                            activation=activation, dropout_probability=dropout_probability,
                                            use_batch_norm=use_batch_norm))
            prev_out_degrees = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(in_degrees=prev_out_degrees, out_features=features * output_multiplier,
                                        autoregressive_features=features, random_mask=random_mask, is_output=True,
                                        out_degrees_=input_degrees_)

    def packed_forward(self, device, spline=False):
        """Device copies of the one-launch pack (flows/made_pack.py), rebuilt when a parameter changes; None = unsupported.
        spline: the final layer in groups for the fused spline epilogue (nf_made_forward_spline)."""
        from . import config
        if not config.made_fused:
            return None
        key = _keys.pkey(self.parameters()) + (str(device),)
        caches = self.__dict__.setdefault("_fwd_pack_cache", {})
        if not isinstance(caches, dict):
            caches = self.__dict__["_fwd_pack_cache"] = {}
        cache = caches.get(bool(spline))
        if cache is None or cache[0] != key:
            from .flows import made_pack
            mult = self.final_layer.out_features // self.initial_layer.in_features
            if not spline and str(device) != "cpu":
                # the value-independent structure (shared by every layer with these masks) + ONE gather on the device: a changed
                # parameter costs microseconds here, not a 0.4 s host repack (the spline pack carries scaled rows: host path below)
                plist = [t for l in self._linears() for t in (l.weight, l.bias)]
                st = _train_struct_for(self, lambda: made_pack.made_train_structure(self, mult), device,
                                       skey=tuple((l.mask.data_ptr(), l.mask._version) for l in self._linears()))
                packed = None
                if st is not None:
                    from . import ops
                    packed = (ops.pack_gather(plist, st["src"][:st["nfwd"]]), st["table"], st["hp"], mult)
            else:
                packed = made_pack.pack_made_forward(self, mult, spline=bool(spline))
                if packed is not None:
                    blob, table = packed
                    packed = (torch.from_numpy(blob).to(device), torch.from_numpy(table).to(device), int(table[3]), mult)
            cache = caches[bool(spline)] = (key, packed)
        return cache[1]

    def _linears(self):
        return [self.initial_layer] + [l for b in self.blocks for l in b.linear_layers] + [self.final_layer]

    def forward(self, inputs, context=None):
        if context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda:
            from . import config
            grad = torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))
            if not grad:
                packed = self.packed_forward(inputs.device)      # nf_made_forward: the whole network as one launch
                if packed is not None:
                    from . import ops
                    return ops.made_forward(inputs, packed[0], packed[1], packed[2], packed[3])
            elif config.made_train and config.made_fused:        # under autograd: hand-written backward (csrc/made_bwd.hip)
                from .flows import made_pack
                plist = [t for l in self._linears() for t in (l.weight, l.bias)]
                mult = self.final_layer.out_features // self.initial_layer.in_features
                packs = _train_packs_from(self, lambda: made_pack.made_train_structure(self, mult), plist, inputs.device,
                                          skey=tuple((l.mask.data_ptr(), l.mask._version) for l in self._linears()))
                if packs is not None:
                    from . import autograd
                    return autograd.MadeFn.apply(packs[0], packs[1], inputs, *plist)
        outputs = self.initial_layer(self.preprocessing(inputs))
        if context is not None:
            outputs = outputs + self.context_layer(context)
        for block in self.blocks:
            outputs = block(outputs, context)
        return self.final_layer(outputs)
