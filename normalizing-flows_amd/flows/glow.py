"""GlowBlock = [AffineCouplingBlock(ConvNet2d), Invertible1x1Conv, ActNorm] (normflows/flows/affine/glow.py:11-84)."""
import torch
from .. import _keys
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
        key = (inverse,) + _keys.pkey(params)
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
        """(packed conditioner, layout, LeakyReLU slope, scale map) when the block runs as one launch for inputs like `z`
        (channel split, scale = True, the 3x3 -> 1x1 -> 3x3 conditioner around 256 channels, float32, enough pixels)."""
        if z.dim() != 4 or z.dtype != torch.float32 or not z.is_cuda:
            return None
        ent = self._level_entry(*z.shape, True)
        return None if ent is None else (ent[0][0], ent[1], ent[2], ent[3])

    def _level_entry(self, B, C, H, W, inverse):
        """What nf_glow_level needs from this block for (B, C, H, W) float32 inputs -- (table entry, layout, slope, scale
        map) -- or None when the block is not the shape the kernels take, its ActNorm is not initialised yet, or gradients
        are required."""
        from ..autograd import needs_grad
        if len(self.flows) != 3 or C < 2 or needs_grad(self):
            return None
        blk = self.flows[0]
        if not isinstance(blk, AffineCouplingBlock) or blk.split_mode != "channel":
            return None
        coupling = blk.flows[1]
        net = coupling.param_map
        if not coupling.scale or not isinstance(net, nets.ConvNet2d):
            return None
        mix = self._fused_mix(inverse)
        if mix is None or mix[0].dtype != torch.float32 or mix[0].shape[0] != C:
            return None
        c1 = (C + 1) // 2
        if net.net[0].in_channels != c1 or net.net[-1].out_channels != 2 * (C - c1):
            return None
        fused = net._fused_pack_for(B, H, W)
        if fused is None:
            return None
        Wp, bp, ldp = mix
        return (fused[0], Wp, bp, ldp), fused[1], net.net[1].negative_slope, coupling.scale_map

    def _run(self, z, inverse, ld, acc, **kw):
        from .. import ops
        from ..autograd import needs_grad
        if len(self.flows) == 3 and z.dim() == 4 and not needs_grad(z, self):
            if inverse:                       # ActNorm's data-dependent init sees the block input first
                self.flows[2]._maybe_init(z, True)
            mix = self._fused_mix(inverse)
            if mix is not None:
                Wp, bp, ldp = mix
                ent = self._level_entry(*z.shape, inverse) if z.dtype == torch.float32 and z.is_cuda else None
                if ent is not None:   # coupling + conditioner + mix: one launch (csrc/glow_conv.hip, nf_glow_level)
                    entry, layout, slope, smap = ent
                    refused = self.__dict__.setdefault("_whole_refused", set())
                    key = (layout, tuple(z.shape[1:]))
                    if key not in refused:
                        try:
                            y, _ = run_level([self], [entry], layout, slope, smap, z, None, False, inverse, ld, acc)
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


_MAX_LEVEL_BLOCKS = 64     # GL_MAXB of csrc/glow_conv.hip


def run_level(blocks, entries, layout, slope, scale_map, in0, in1, in_squeezed, inverse, ld, acc, cout0=None,
              out_squeezed=False):
    """`blocks` (GlowBlocks in PROCESSING order, with their _level_entry table entries) as one persistent launch.  The
    device pointer table is cached on the first block per (direction, member tensors): building it is a host -> device copy,
    and a recorded hipGraph bakes its address in."""
    from .. import ops
    first = blocks[0]
    key = (inverse, layout) + tuple(t.data_ptr() for e in entries for t in e)
    cache = first.__dict__.setdefault("_level_tbl_cache", {})
    hit = cache.get((inverse, len(blocks)))
    if hit is None or hit[0] != key:
        ents = [(b_, w_.contiguous(), bb_.contiguous(), l_.to(torch.float32).contiguous()) for b_, w_, bb_, l_ in entries]
        hit = cache[(inverse, len(blocks))] = (key,) + ops.glow_block_table(ents, in0.device)
    if in_squeezed:
        C, H, W = in0.shape[1] * 4, in0.shape[2] // 2, in0.shape[3] // 2
    else:
        C, H, W = in0.shape[1] + (0 if in1 is None else in1.shape[1]), in0.shape[2], in0.shape[3]
    out0, out1, _ = ops.glow_level(in0, in1, in_squeezed, C, H, W, hit[1], len(blocks), layout, slope, scale_map,
                                   1 if inverse else 0, cout0=cout0, out_squeezed=out_squeezed, logdet=ld, acc=acc)
    return out0, out1


def plan_level(flows, B, C, H, W, inverse):
    """Longest prefix of `flows` (GlowBlocks in processing order, <= 64) that nf_glow_level takes as ONE launch on
    (B, C, H, W) float32 inputs: (n, entries, layout, slope, scale_map); n = 0 when the first block is not eligible."""
    entries, sig = [], None
    for f in flows[:_MAX_LEVEL_BLOCKS]:
        if not isinstance(f, GlowBlock):
            break
        ent = f._level_entry(B, C, H, W, inverse)
        if ent is None or (sig is not None and ent[1:] != sig):
            break
        sig = ent[1:]
        entries.append(ent[0])
    if not entries:
        return 0, None, None, None, None
    return (len(entries), entries) + sig
