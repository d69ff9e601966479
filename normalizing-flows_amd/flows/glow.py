"""GlowBlock = [AffineCouplingBlock(ConvNet2d), Invertible1x1Conv, ActNorm] (normflows/flows/affine/glow.py:11-84)."""
import torch
from torch import nn

from .. import nets
from .affine import AffineCouplingBlock
from .base import Flow, run_flow
from .mixing import Invertible1x1Conv
from .normalization import ActNorm


class GlowBlock(Flow):
    def __init__(self, channels, hidden_channels, scale=True, scale_map="sigmoid", split_mode="channel", leaky=0.0,
                 init_zeros=True, use_lu=True, net_actnorm=False):
        super().__init__()
        self.flows = nn.ModuleList([])
        kernel_size = (3, 1, 3)
        num_param = 2 if scale else 1
        if "channel" == split_mode:
            channels_ = ((channels + 1) // 2,) + 2 * (hidden_channels,)
            channels_ += (num_param * (channels // 2),)
        elif "channel_inv" == split_mode:
            channels_ = (channels // 2,) + 2 * (hidden_channels,)
            channels_ += (num_param * ((channels + 1) // 2),)
        elif "checkerboard" in split_mode:
            channels_ = (channels,) + 2 * (hidden_channels,)
            channels_ += (num_param * channels,)
        else:
            raise NotImplementedError("Mode " + split_mode + " is not implemented.")
        param_map = nets.ConvNet2d(channels_, kernel_size, leaky, init_zeros, actnorm=net_actnorm)
        self.flows += [AffineCouplingBlock(param_map, scale, scale_map, split_mode)]
        if channels > 1:
            self.flows += [Invertible1x1Conv(channels, use_lu)]
        self.flows += [ActNorm((channels,) + (1, 1))]

    def _fused_mix(self, inverse):
        """[Invertible1x1Conv, ActNorm] as ONE per-pixel affine map y = W' z + b' (inference, parameters frozen between
        updates): inverse  y = W (z - t) e^-s  ->  W' = W diag(e^-s), b' = -W' t, log|det| per pixel = ld_W - sum s;
        forward  y = (W^-1 z) e^s + t  ->  W' = diag(e^s) W^-1, b' = t, log|det| = ld_Winv + sum s."""
        conv, an = self.flows[1], self.flows[2]
        if not (isinstance(conv, Invertible1x1Conv) and isinstance(an, ActNorm) and an._per_channel):
            return None
        if an._init_known is None:
            an._init_known = bool(an.data_dep_init_done.item() > 0.0)
        if not an._init_known:
            return None
        params = [an.s, an.t] + ([conv.L, conv.U, conv.log_S] if conv.use_lu else [conv.W])
        key = (inverse,) + tuple((p_.data_ptr(), p_._version) for p_ in params)
        cache = getattr(self, "_mix_cache", None)
        if cache is None or cache[0] != key:
            W, ldu = conv._weight(inverse)
            s_, t_ = an.s.detach().reshape(-1), an.t.detach().reshape(-1)
            if inverse:
                Wp = W * torch.exp(-s_)[None, :]
                bp = -(Wp @ t_)
                ldp = ldu - s_.sum()
            else:
                Wp = torch.exp(s_)[:, None] * W
                bp = t_.clone()
                ldp = ldu + s_.sum()
            cache = (key, (Wp.contiguous(), bp.contiguous(), ldp))
            self._mix_cache = cache
        return cache[1]

    def _whole_block(self, z):
        """(packed conditioner, layout, LeakyReLU slope, scale map) when the block is the shape nf_glow_block takes:
        channel split, scale = True, the 3x3 -> 1x1 -> 3x3 conditioner around 256 channels, float32, enough pixels."""
        blk = self.flows[0]
        if not isinstance(blk, AffineCouplingBlock) or blk.split_mode != "channel" or z.shape[1] < 2:
            return None
        coupling = blk.flows[1]
        net = coupling.param_map
        if not coupling.scale or not isinstance(net, nets.ConvNet2d) or z.dtype != torch.float32:
            return None
        zc = z if z.is_contiguous() else z.contiguous()
        c1 = (z.shape[1] + 1) // 2
        fused = net._fused_pack(zc[:, :c1])
        if fused is None or net.net[-1].out_channels != 2 * (z.shape[1] - c1):
            return None
        return fused[0], fused[1], net.net[1].negative_slope, coupling.scale_map

    def _run(self, z, inverse, ld, acc, **kw):
        from .. import ops
        from ..autograd import needs_grad
        if len(self.flows) == 3 and z.dim() == 4 and not needs_grad(z, self):
            if inverse:                       # ActNorm's data-dependent init sees the block input first
                self.flows[2]._maybe_init(z, True)
            mix = self._fused_mix(inverse)
            if mix is not None:
                Wp, bp, ldp = mix
                whole = self._whole_block(z)
                if whole is not None:   # coupling + conditioner + mix: one launch (csrc/glow_conv.hip, nf_glow_block)
                    blob, layout, slope, smap = whole
                    refused = self.__dict__.setdefault("_whole_refused", set())
                    key = (layout, tuple(z.shape[1:]))
                    if key not in refused:
                        try:
                            y, _ = ops.glow_block(z, blob, layout, Wp, bp, ldp, slope, smap, 1 if inverse else 0,
                                                  logdet=ld, acc=acc)
                            return y
                        except NotImplementedError:   # the block's working set does not fit one workgroup's LDS:
                            refused.add(key)          # nothing was launched; keep the layer-by-layer path for this shape
                if inverse:
                    z, _ = ops.inv1x1_conv(z, Wp, ldp, logdet=ld, acc=acc, want_scalar=False, bias=bp)
                    return run_flow(self.flows[0], z, True, ld, acc)
                z = run_flow(self.flows[0], z, False, ld, acc)
                z, _ = ops.inv1x1_conv(z, Wp, ldp, logdet=ld, acc=acc, want_scalar=False, bias=bp)
                return z
        seq = reversed(self.flows) if inverse else self.flows
        for f in seq:
            z = run_flow(f, z, inverse, ld, acc)
        return z

    def forward(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._run(z, False, ld, +1), ld

    def inverse(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._run(z, True, ld, +1), ld
