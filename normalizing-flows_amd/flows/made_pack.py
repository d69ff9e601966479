"""Host-side packing for the one-launch MADE forward kernel (nf_made_forward_affine / nf_made_forward, csrc/made_fwd.hip).

The single-pass direction of the autoregressive flows (normflows/flows/affine/autoregressive.py:24-27: `forward` = ONE pass of
nets/made.py:296-304 + the element-wise transform) is a stack of MaskedLinear layers (made.py:19-81: `F.linear(x, weight * mask,
bias)`).  With the hidden units SORTED BY DEGREE every mask is block lower-triangular -- a unit of degree m only sees inputs of
degree <= m (:63-81) -- so about half of the 32 x 8 weight blocks are structurally zero and are neither stored nor multiplied.

This module only rearranges weights (no arithmetic on data).  The kernel's geometry (made_fwd.hip):
  * hidden slots = units sorted by degree (stable), zero-padded to Hp = 256 or 512; row-block rb = slots [32 rb, 32 rb + 32);
  * a k-group kg = 8 consecutive inputs (features for the initial layer, hidden slots otherwise); features padded to Dp (x 32);
  * a work item = one row-block of one layer: a BIAS group (4 KB: entry q, every lane of half hh = bias[32 rb + 8 q + 4 hh + 0..3])
    followed by nkg A-operand fragments of 1 KB: lane (hh = lane >> 5, m = lane & 31) holds W[32 rb + m][8 kg + 4 hh + 0..3]
    (v_mfma_f32_32x32x2_f32: A[m][k = hh], four k-steps per 16-byte load); nkg = the last k-group in which the MASK of the
    row-block has a non-zero, + 1, rounded up to 4 (a prefix, by the sorting);
  * each of the kernel's 8 waves walks ONE contiguous stream: its items in consumption order -- per producing layer (initial,
    the residual blocks' linears) the row-blocks {w, HRB - 1 - w} (HRB = Hp / 32: equal work per wave, see work_per_wave), per
    round r of the final layer the row-blocks {8 r + w, 8 r + 7 - w} -- followed by a copy of its first RING entries, so that the
    register ring that reads RING entries ahead wraps into the next 64-row tile without a bubble.

  * `spline=True` (the autoregressive spline layer's MADE, 23 = 3 * 8 - 1 outputs per feature; nf_made_forward_spline): the final
    layer is cut into GROUPS of four features = 3 row-blocks for both sample blocks, rows ordered so that a lane's accumulators are
    the 2 x 24 parameter lists of features 4 g + 2 hh + {0, 1} (flows/nsf_wide_pack.final_row: the layout of nf_nsf_wide), widths /
    heights pre-scaled by log2(e); a group's stream = 12 bias entries, then per k-group one fragment per row-block; the groups are
    dealt to the waves in a snake over their k-group counts (equal work); hdr[8] = final items per wave, entries [nkg, g].

float blob  : the 8 streams.
int32 table : hdr[32] = [D, Dp, H, Hp, NSB, NB, mult, NFB | G, nrounds | nfi, total blob floats, nitems, spline, 0..], hdr[16 + w] =
              offset (floats) of wave w's stream; then per wave nitems entries [nkg, rb | g] (-1: none, nkg = 0, no bias group).
"""
import numpy as np
import torch
from torch.nn import functional as F

HDR = 32
ROWS = 32        # MFMA rows per row-block
KG = 8           # inputs per k-group
MAX_D = 128      # features: the x tile of 64 rows is 32 KB of LDS
MAX_LAYERS = 16
RING = 8         # made_fwd.hip MF_PF: k-groups of A a wave keeps in flight


def supported(made, mult):
    from .. import nets
    if not isinstance(made, nets.MADE):
        return False
    if not isinstance(made.preprocessing, torch.nn.Identity) or hasattr(made, "context_layer"):
        return False
    nb = len(made.blocks)
    if nb < 1 or 2 * nb + 2 > MAX_LAYERS or not all(isinstance(b, nets.MaskedResidualBlock) for b in made.blocks):
        return False
    for b in made.blocks:
        if b.use_batch_norm or b.activation is not F.relu or b.dropout.p != 0.0 or hasattr(b, "context_layer"):
            return False
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    if made.final_layer.out_features != mult * D or made.initial_layer.weight.dtype != torch.float32:
        return False
    return 2 <= D <= MAX_D and 1 <= H <= 512 and mult >= 1


def a_stream(w_block):
    """(32, 8 nkg) row-major -> [nkg][2][32][4] (lane = 32 hh + m holds W[m][8 kg + 4 hh + 0..3])."""
    rows, K = w_block.shape
    assert rows == ROWS and K % KG == 0
    return np.ascontiguousarray(w_block.reshape(ROWS, K // KG, 2, 4).transpose(1, 2, 0, 3)).reshape(-1)


def bias_group(b32):
    """32 biases of a row-block -> [4 q][2 hh][32 m][4]: entry q, lane (hh, m) = b[8 q + 4 hh + 0..3] (accumulator register 4 q + i)."""
    return np.ascontiguousarray(np.broadcast_to(b32.reshape(4, 2, 1, 4), (4, 2, ROWS, 4))).reshape(-1)


def wave_items(NSB, NB, NFB):
    """Row-blocks of wave w in consumption order: [(layer, rb or -1), ...] (layer 2 NB + 1 = final)."""
    HRB = 8 * NSB
    nrounds = (NFB + 7) // 8
    out = []
    for w in range(8):
        it = []
        for l in range(1 + 2 * NB):
            it += [(l, w), (l, HRB - 1 - w)]
        for r in range(nrounds):
            for fb in (8 * r + w, 8 * r + 7 - w):
                it.append((2 * NB + 1, fb if fb < NFB else -1))
        out.append(it)
    return out


def _slot_layers(made, mult, wb=None):
    """The MADE's masked linears in SLOT space (hidden units sorted by degree, zero-padded): dict with the sizes, `order` (slot ->
    unit) and `layers` = [(W, M, b)] for the initial layer, the blocks' linears and the final layer; None outside the structure.
    wb: [(weight, bias)] arrays to rearrange instead of the module's parameters (index_arrays: the packs as gather indices)."""
    if not supported(made, mult):
        return None
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    NB = len(made.blocks)
    hid_deg = made.initial_layer.degrees.cpu().numpy()
    lin = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers]
    for l in lin[1:]:
        if not np.array_equal(l.degrees.cpu().numpy(), hid_deg):
            return None
    fin = made.final_layer
    # inputs in degree order (no permute_mask) and the output rows feature-major: what the element-wise epilogue assumes
    m0 = made.initial_layer.mask.cpu().numpy()
    if not np.array_equal(m0, (hid_deg[:, None] >= np.arange(1, D + 1)[None, :]).astype(m0.dtype)):
        return None
    mf = fin.mask.cpu().numpy()
    out_deg = np.repeat(np.arange(1, D + 1), mult)
    if not np.array_equal(mf, (out_deg[:, None] > hid_deg[None, :]).astype(mf.dtype)):
        return None
    Hp = 256 if H <= 256 else 512
    NSB = Hp // 256
    Dp = (D + 31) // 32 * 32
    NFB = (mult * D + ROWS - 1) // ROWS
    order = np.argsort(hid_deg, kind="stable")            # slot i holds unit order[i]
    slot_of = np.zeros(H, dtype=np.int64)
    slot_of[order] = np.arange(H)

    all_lin = lin + [fin]

    def slots(lyr, in_map, in_size, out_rows, out_size):
        """Masked weight, mask and bias of `lyr` in slot space (rows -> out_rows, columns -> in_map)."""
        m = lyr.mask.cpu().numpy() != 0
        if wb is None:
            w = (lyr.weight.detach() * lyr.mask).cpu().numpy().astype(np.float32)
            b = lyr.bias.detach().cpu().numpy().astype(np.float32)
        else:
            k = [id(x) for x in all_lin].index(id(lyr))
            w = (np.asarray(wb[k][0]) * m).astype(np.float32)
            b = np.asarray(wb[k][1]).astype(np.float32)
        W = np.zeros((out_size, in_size), dtype=np.float32)
        M = np.zeros((out_size, in_size), dtype=bool)
        Bv = np.zeros(out_size, dtype=np.float32)
        tmp = np.zeros((w.shape[0], in_size), dtype=np.float32)
        tmp[:, in_map] = w
        W[out_rows] = tmp
        tm = np.zeros((w.shape[0], in_size), dtype=bool)
        tm[:, in_map] = m
        M[out_rows] = tm
        Bv[out_rows] = b
        return W, M, Bv

    layers = [slots(lin[0], np.arange(D), Dp, slot_of, Hp)]
    for l in lin[1:]:
        layers.append(slots(l, slot_of, Hp, slot_of, Hp))
    layers.append(slots(fin, slot_of, Hp, np.arange(mult * D), NFB * ROWS))
    return dict(D=D, H=H, NB=NB, Hp=Hp, NSB=NSB, Dp=Dp, NFB=NFB, MD=mult * D, mult=mult, order=order, slot_of=slot_of,
                layers=layers, masks=[l.mask.cpu().numpy() != 0 for l in lin + [fin]])


def resnet_supported(net):
    from .. import nets
    if not isinstance(net, nets.ResidualNet) or not net.is_plain_relu():
        return False
    if len(net.blocks) < 1 or 2 * len(net.blocks) + 2 > MAX_LAYERS:
        return False
    D, H, MD = net.initial_layer.in_features, net.initial_layer.out_features, net.final_layer.out_features
    return 2 <= D <= MAX_D and 1 <= H <= 512 and MD >= 1 and net.initial_layer.weight.dtype == torch.float32


def _dense_layers(net, wb=None):
    """A ResidualNet (nets/resnet.py:53-104: the same layer structure as MADE without masks) as slot-space layers for the MADE
    kernels: identity unit order, every mask all ones (no block is skipped), out_features = MD.  wb: as in _slot_layers."""
    if not resnet_supported(net):
        return None
    D, H, MD = net.initial_layer.in_features, net.initial_layer.out_features, net.final_layer.out_features
    NB = len(net.blocks)
    Hp = 256 if H <= 256 else 512
    Dp = (D + 31) // 32 * 32
    NFB = (MD + ROWS - 1) // ROWS
    lin = [net.initial_layer] + [l for b in net.blocks for l in b.linear_layers] + [net.final_layer]
    sizes = [(Hp, Dp)] + [(Hp, Hp)] * (2 * NB) + [(NFB * ROWS, Hp)]
    layers = []
    for k, (l, (o, i)) in enumerate(zip(lin, sizes)):
        w = (l.weight.detach().cpu().numpy() if wb is None else np.asarray(wb[k][0])).astype(np.float32)
        W = np.zeros((o, i), dtype=np.float32)
        M = np.zeros((o, i), dtype=bool)
        Bv = np.zeros(o, dtype=np.float32)
        W[:w.shape[0], :w.shape[1]] = w
        M[:w.shape[0], :w.shape[1]] = True
        Bv[:w.shape[0]] = (l.bias.detach().cpu().numpy() if wb is None else np.asarray(wb[k][1])).astype(np.float32)
        layers.append((W, M, Bv))
    return dict(D=D, H=H, NB=NB, Hp=Hp, NSB=Hp // 256, Dp=Dp, NFB=NFB, MD=MD, mult=1, order=np.arange(H), slot_of=np.arange(H),
                layers=layers, masks=[np.ones(tuple(l.weight.shape), dtype=bool) for l in lin])


def pack_made_forward(made, mult=2, spline=False):
    """(blob float32 ndarray, table int32 ndarray) or None when the MADE is outside the kernel's structure (then the caller
    keeps the layer-by-layer path).  spline: the final layer in groups for the fused spline epilogue (mult = 23)."""
    if spline and mult != 23:
        return None
    sl = _slot_layers(made, mult)
    if sl is None:
        return None
    if spline:
        D, H, NB, Hp, NSB, Dp, layers = (sl[k] for k in ("D", "H", "NB", "Hp", "NSB", "Dp", "layers"))
        return _pack_spline(layers, D, Dp, H, Hp, NSB, NB, mult)
    return _pack_forward_from(sl)


def pack_resnet_forward(net):
    """The same blob / table for a ResidualNet (dense: no block skipped), hdr[12] = out_features; None outside the structure."""
    sl = _dense_layers(net)
    return None if sl is None else _pack_forward_from(sl)


def _pack_forward_from(sl):
    D, H, NB, Hp, NSB, Dp, NFB, layers, mult = (sl[k] for k in ("D", "H", "NB", "Hp", "NSB", "Dp", "NFB", "layers", "mult"))
    items = wave_items(NSB, NB, NFB)
    nitems = len(items[0])
    hdr = np.zeros(HDR, dtype=np.int32)
    tab = np.zeros((8, nitems, 2), dtype=np.int32)
    chunks = []
    off = 0
    for w in range(8):
        hdr[16 + w] = off
        stream = []
        for i, (l, rb) in enumerate(items[w]):
            if rb < 0 or layers[l] is None:          # (None: the block's second linear in plain-MLP mode)
                tab[w, i] = (0, -1)
                continue
            W, M, Bv = layers[l]
            r0 = rb * ROWS
            cols = np.nonzero(M[r0:r0 + ROWS].any(axis=0))[0]
            nkg = 0 if cols.size == 0 else (int(cols.max()) // KG + 4) // 4 * 4        # + 1, rounded up to 4
            assert KG * nkg <= W.shape[1]
            tab[w, i] = (nkg, rb)
            stream.append(bias_group(Bv[r0:r0 + ROWS]))
            if nkg:
                stream.append(a_stream(W[r0:r0 + ROWS, :KG * nkg]))
        stream = np.concatenate(stream)
        stream = np.concatenate([stream, np.resize(stream, RING * 256)])      # the ring wraps into the next tile
        chunks.append(stream)
        off += stream.size
    hdr[:11] = [D, Dp, H, Hp, NSB, NB, mult, NFB, (NFB + 7) // 8, off, nitems]
    hdr[12] = sl["MD"]                                      # output row length (mult D for a MADE)
    hdr[13] = 1 if sl.get("plain") else 0
    blob = np.concatenate(chunks).astype(np.float32)
    assert blob.size == off and off < 2 ** 31
    return blob, np.concatenate([hdr, tab.reshape(-1)]).astype(np.int32)


def _pack_spline(layers, D, Dp, H, Hp, NSB, NB, mult):
    """Streams with the final layer in groups of four features (see the module docstring)."""
    from .nsf_wide_pack import final_row, K_BINS, M as MPRM
    Wf, Mf, bf = layers[-1]                                   # (NFB * 32, Hp) in the reference's row order 23 f + p
    G = (D + 3) // 4
    scale = np.float32(1.4426950408889634)                    # log2(e): rqs_regs takes exp2; no 1 / sqrt(hidden) here (the reference
    WG = np.zeros((G, 3, ROWS, Hp), dtype=np.float32)         # tests hasattr(net, "hidden_features"), which its MADE does not have)
    BG = np.zeros((G, 3, ROWS), dtype=np.float32)
    nkg_g = np.zeros(G, dtype=np.int64)
    for g in range(G):
        last = -1
        for r3 in range(3):
            for rho in range(ROWS):
                row = final_row(g, r3, rho, D)
                if row >= 0:
                    sc = scale if (row % MPRM) < 2 * K_BINS else np.float32(1.0)
                    WG[g, r3, rho] = Wf[row] * sc
                    BG[g, r3, rho] = bf[row] * sc
                    nz = np.nonzero(Mf[row])[0]
                    if nz.size:
                        last = max(last, int(nz.max()))
        nkg_g[g] = 0 if last < 0 else (last // KG + 4) // 4 * 4
    # deal the groups to the waves in a snake over their k-group counts
    order = np.argsort(-nkg_g, kind="stable")
    per_wave = [[] for _ in range(8)]
    for k, g in enumerate(order):
        r, c = divmod(k, 8)
        per_wave[c if r % 2 == 0 else 7 - c].append(int(g))
    nfi = max(len(v) for v in per_wave)
    hidden = wave_items(NSB, NB, 0)                           # the producing layers' items (no final rounds)
    nh = len(hidden[0])
    nitems = nh + nfi
    hdr = np.zeros(HDR, dtype=np.int32)
    tab = np.zeros((8, nitems, 2), dtype=np.int32)
    chunks, off = [], 0
    for w in range(8):
        hdr[16 + w] = off
        stream = []
        for i, (l, rb) in enumerate(hidden[w]):
            W, M, Bv = layers[l]
            r0 = rb * ROWS
            cols = np.nonzero(M[r0:r0 + ROWS].any(axis=0))[0]
            nkg = 0 if cols.size == 0 else (int(cols.max()) // KG + 4) // 4 * 4
            tab[w, i] = (nkg, rb)
            stream.append(bias_group(Bv[r0:r0 + ROWS]))
            if nkg:
                stream.append(a_stream(W[r0:r0 + ROWS, :KG * nkg]))
        for j in range(nfi):
            if j >= len(per_wave[w]):
                tab[w, nh + j] = (0, -1)
                continue
            g = per_wave[w][j]
            nkg = int(nkg_g[g])
            tab[w, nh + j] = (nkg, g)
            for r3 in range(3):
                stream.append(bias_group(BG[g, r3]))
            if nkg:
                frag = np.stack([a_stream(WG[g, r3][:, :KG * nkg]).reshape(nkg, 256) for r3 in range(3)], axis=1)
                stream.append(frag.reshape(-1))
        stream = np.concatenate(stream)
        stream = np.concatenate([stream, np.resize(stream, RING * 256)])
        chunks.append(stream)
        off += stream.size
    hdr[:12] = [D, Dp, H, Hp, NSB, NB, mult, G, nfi, off, nitems, 1]
    blob = np.concatenate(chunks).astype(np.float32)
    assert blob.size == off and off < 2 ** 31
    return blob, np.concatenate([hdr, tab.reshape(-1)]).astype(np.int32)


# ---- backward pass (csrc/made_bwd.hip) ------------------------------------------------------------------------------------------------
W_TILE = 128     # made_wgrad_kernel: edge of an output tile


def _suffix_item(WT, MT, rb, c0, c1):
    """Row-block rb of a transposed masked weight restricted to the columns [c0, c1): (nkg, kg0 relative to c0, stream) -- the
    k-groups between the first and the last mask non-zero, both rounded to multiples of 4 (a suffix once the units are sorted)."""
    r0 = rb * ROWS
    cols = np.nonzero(MT[r0:r0 + ROWS, c0:c1].any(axis=0))[0]
    zero_bias = np.zeros(4 * 256, dtype=np.float32)
    if cols.size == 0:
        return 0, 0, [zero_bias]
    kg0 = (int(cols.min()) // KG) // 4 * 4
    kg1 = (int(cols.max()) // KG + 4) // 4 * 4
    assert KG * kg1 <= c1 - c0
    return kg1 - kg0, kg0, [zero_bias, a_stream(WT[r0:r0 + ROWS, c0 + KG * kg0:c0 + KG * kg1])]


def pack_made_backward(made, mult=2):
    """Tables of nf_made_backward / nf_made_wgrad for a MADE pack_made_forward takes, or None.  Returns a dict of numpy arrays:
      blob, table        the input-gradient chain's 8 streams (zero bias group + the TRANSPOSED masked weight's fragments per item) and
                         hdr[32] = [D, Dp, H, Hp, NSB, NB, mult, NC (chunks of Hp columns of g_params), 0, total, nitems], hdr[16 + w]
                         = stream offsets, then per wave nitems entries [nkg, rb, kg0, 0]: NC x 2 items of Wf^T, per block (last
                         first) 2 of W2^T and 2 of W1^T, one of W0^T (row-block w & 3 of the features for sample block w >> 2, -1: none)
      wtable             hdr[16] = [ntiles, nproblems], problems [dY base, dY matrix, ldY, X base, X matrix, ldX, relu, 0], tiles
                         [problem, m0, n0, want_bias, 0...] (bases: 0 = g_params padded, 1 = x padded, 2 = G, 3 = save)
      stable             per problem [weight offset, ld, bias offset, row-map offset, column-map offset, 0, 0, 0], then the maps
      mask               bytes over the flat gradient layout (1 where the parameter's mask is non-zero; biases 1)
      offsets            [(weight offset, shape, bias offset, n)] per linear in the order initial, blocks' linears, final
      nflat, ntiles, Mp (g_params row length padded to 128), Dx (x row length padded to 128)."""
    sl = _slot_layers(made, mult)
    return None if sl is None else _pack_backward_from(sl)


def pack_resnet_backward(net):
    """pack_made_backward for a ResidualNet (all masks ones)."""
    sl = _dense_layers(net)
    return None if sl is None else _pack_backward_from(sl)


def _pack_backward_from(sl):
    D, H, NB, Hp, NSB, Dp, NFB, layers, order, mult, MD = (sl[k] for k in ("D", "H", "NB", "Hp", "NSB", "Dp", "NFB", "layers", "order",
                                                                           "mult", "MD"))
    HRB = Hp // ROWS
    NC = (MD + Hp - 1) // Hp
    Wf, Mf, _ = layers[-1]
    WfT = np.zeros((Hp, NC * Hp), dtype=np.float32)
    MfT = np.zeros((Hp, NC * Hp), dtype=bool)
    WfT[:, :Wf.shape[0]] = Wf.T
    MfT[:, :Mf.shape[0]] = Mf.T
    plain = bool(sl.get("plain"))
    nfin = (Dp // ROWS + 3) // 4                          # rounds of the last product (4 feature row-blocks x 2 sample blocks each)
    nitems = 2 * NC + 4 * NB + nfin
    hdr = np.zeros(HDR, dtype=np.int32)
    tab = np.zeros((8, nitems, 4), dtype=np.int32)
    chunks, off = [], 0
    for w in range(8):
        hdr[16 + w] = off
        stream, i = [], 0
        for c in range(NC):
            for rb in (w, HRB - 1 - w):
                nkg, kg0, st = _suffix_item(WfT, MfT, rb, c * Hp, (c + 1) * Hp)
                tab[w, i] = (nkg, rb, kg0, 0)
                stream += st
                i += 1
        for b in range(NB - 1, -1, -1):
            for l in (2 + 2 * b, 1 + 2 * b):             # W2 of the block, then its W1
                for rb in (w, HRB - 1 - w):
                    if layers[l] is None:                # (plain-MLP mode: no second linear)
                        tab[w, i] = (0, -1, 0, 0)
                    else:
                        nkg, kg0, st = _suffix_item(layers[l][0].T, layers[l][1].T, rb, 0, Hp)
                        tab[w, i] = (nkg, rb, kg0, 0)
                        stream += st
                    i += 1
        W0, M0, _ = layers[0]
        for rd in range(nfin):
            rb = (w & 3) + 4 * rd
            if rb < Dp // ROWS:
                nkg, kg0, st = _suffix_item(W0.T, M0.T, rb, 0, Hp)
                tab[w, i] = (nkg, rb, kg0, 0)
                stream += st
            else:
                tab[w, i] = (0, -1, 0, 0)
            i += 1
        stream = np.concatenate(stream)
        stream = np.concatenate([stream, np.resize(stream, RING * 256)])
        chunks.append(stream)
        off += stream.size
    hdr[:11] = [D, Dp, H, Hp, NSB, NB, mult, NC, nfin, off, nitems]
    hdr[12] = MD
    hdr[13] = 1 if plain else 0
    blob = np.concatenate(chunks).astype(np.float32)
    assert blob.size == off and off < 2 ** 31
    table = np.concatenate([hdr, tab.reshape(-1)]).astype(np.int32)

    # ---- weight gradients: problems, non-zero 128 x 128 tiles, scatter maps, flat layout ----
    Mp = (MD + W_TILE - 1) // W_TILE * W_TILE
    Dx = (D + W_TILE - 1) // W_TILE * W_TILE
    slot_to_unit = np.full(Hp, -1, dtype=np.int32)
    slot_to_unit[:H] = order
    feat_map = np.full(Dx, -1, dtype=np.int32)
    feat_map[:D] = np.arange(D)
    out_map = np.full(Mp, -1, dtype=np.int32)
    out_map[:MD] = np.arange(MD)
    # (problem: dY base / matrix / ld, X base / matrix / ld, relu, slot-space mask [M x N], row map, column map, parameter shape)
    probs = [(2, 0, Hp, 1, 0, Dx, 0, layers[0][1], slot_to_unit, feat_map, (H, D))]
    if plain:      # x -> W0 -> relu -> W1 -> relu -> Wf: dW1 = G[1]^T relu(save[0]), dWf = g_p^T relu(save[1])
        probs.append((2, 1, Hp, 3, 0, Hp, 1, layers[1][1], slot_to_unit, slot_to_unit, (H, H)))
        probs.append((0, 0, Mp, 3, 1, Hp, 1, layers[-1][1], out_map, slot_to_unit, (MD, H)))
    else:
        for b in range(NB):
            probs.append((2, 2 * b + 1, Hp, 3, 2 * b, Hp, 1, layers[1 + 2 * b][1], slot_to_unit, slot_to_unit, (H, H)))
            probs.append((2, 2 * b + 2, Hp, 3, 2 * b + 1, Hp, 1, layers[2 + 2 * b][1], slot_to_unit, slot_to_unit, (H, H)))
        probs.append((0, 0, Mp, 3, 2 * NB, Hp, 0, layers[-1][1], out_map, slot_to_unit, (MD, H)))
    ptab, tiles, stab, maps = [], [], [], []
    offsets, mask_parts, flat = [], [], 0
    map_off = 8 * len(probs)
    wmaps = sl.get("wmaps")          # per problem (row map, column map, ld, bias map): destinations in the parameter's OWN layout
    for pi, (dyb, dyl, ldy, xb, xl, ldx, relu, M, rmap, cmap, shape) in enumerate(probs):
        ptab.append([dyb, dyl, ldy, xb, xl, ldx, relu, 0])
        Mrows, Ncols = len(rmap), len(cmap)
        ldw, bmap = shape[1], None
        if wmaps is not None:
            r2, c2, ldw, b2 = wmaps[pi]
            bmap = np.full(Mrows, -1, dtype=np.int32)
            bmap[:len(b2)] = b2
            rmap, cmap = np.full(Mrows, -1, dtype=np.int32), np.full(Ncols, -1, dtype=np.int32)
            rmap[:len(r2)] = r2
            cmap[:len(c2)] = c2
        Mfull = np.zeros((Mrows, Ncols), dtype=bool)
        Mfull[:M.shape[0], :M.shape[1]] = M
        for mt in range(Mrows // W_TILE):
            nz = [nt for nt in range(Ncols // W_TILE) if Mfull[mt * W_TILE:(mt + 1) * W_TILE, nt * W_TILE:(nt + 1) * W_TILE].any()]
            if not nz:
                nz = [0]                                  # (the bias gradient still needs the tile row)
            for k, nt in enumerate(nz):
                tiles.append([pi, mt * W_TILE, nt * W_TILE, 1 if k == 0 else 0, 0, 0, 0, 0])
        woff, boff = flat, flat + shape[0] * shape[1]
        flat = boff + shape[0]
        offsets.append((woff, shape, boff, shape[0]))
        stab.append([woff, ldw, boff, map_off, map_off + Mrows, 0 if bmap is None else map_off + Mrows + Ncols, 0, 0])
        maps += [np.asarray(rmap, dtype=np.int32), np.asarray(cmap, dtype=np.int32)]
        map_off += Mrows + Ncols
        if bmap is not None:
            maps.append(bmap)
            map_off += Mrows
        mask_parts += [sl["masks"][pi].astype(np.uint8).reshape(-1), np.ones(shape[0], dtype=np.uint8)]
    whdr = np.zeros(16, dtype=np.int32)
    whdr[:2] = [len(tiles), len(probs)]
    wtable = np.concatenate([whdr, np.asarray(ptab, dtype=np.int32).reshape(-1), np.asarray(tiles, dtype=np.int32).reshape(-1)])
    stable = np.concatenate([np.asarray(stab, dtype=np.int32).reshape(-1)] + maps).astype(np.int32)
    mask = np.concatenate(mask_parts)
    assert mask.size == flat and flat < 2 ** 31
    return dict(blob=blob, table=table, wtable=wtable.astype(np.int32), stable=stable, mask=mask, offsets=offsets, nflat=flat,
                ntiles=len(tiles), Mp=Mp, Dx=Dx, Hp=Hp, NB=NB, mult=mult, MD=MD)


# ---- plain MLP  x -> W0 -> relu -> W1 -> relu -> Wf  (the conv conditioner of GlowBlock over pixel rows, csrc/conv_rows.hip) -----------
def _mlp_layers(W0, b0, W1, b1, Wf, bf):
    """Slot-space layers of a two-hidden-layer ReLU MLP for the MADE kernels' plain mode (hdr[13] = 1): NB = 1 with the block's
    second linear absent; D <= 256 inputs when the hidden width fits 256 slots (128 otherwise)."""
    W0, W1, Wf = (np.asarray(a, dtype=np.float32) for a in (W0, W1, Wf))
    H, D = W0.shape
    MD = Wf.shape[0]
    if W1.shape != (H, H) or Wf.shape[1] != H or not (1 <= H <= 512 and MD >= 1):
        return None
    Hp = 256 if H <= 256 else 512
    if not (2 <= D <= (256 if Hp == 256 else MAX_D)):
        return None
    Dp = (D + 31) // 32 * 32
    NFB = (MD + ROWS - 1) // ROWS

    def pad(w, b, o, i):
        W = np.zeros((o, i), dtype=np.float32)
        M = np.zeros((o, i), dtype=bool)
        Bv = np.zeros(o, dtype=np.float32)
        W[:w.shape[0], :w.shape[1]] = w
        M[:w.shape[0], :w.shape[1]] = True
        if b is not None:
            Bv[:w.shape[0]] = np.asarray(b, dtype=np.float32)
        return W, M, Bv
    layers = [pad(W0, b0, Hp, Dp), pad(W1, b1, Hp, Hp), None, pad(Wf, bf, NFB * ROWS, Hp)]
    return dict(D=D, H=H, NB=1, Hp=Hp, NSB=Hp // 256, Dp=Dp, NFB=NFB, MD=MD, mult=1, order=np.arange(H), slot_of=np.arange(H),
                layers=layers, masks=[np.ones(W0.shape, dtype=bool), np.ones(W1.shape, dtype=bool), np.ones(Wf.shape, dtype=bool)],
                plain=True)


def pack_mlp_forward(W0, b0, W1, b1, Wf, bf=None):
    sl = _mlp_layers(W0, b0, W1, b1, Wf, bf)
    return None if sl is None else _pack_forward_from(sl)


def pack_mlp_backward(W0, b0, W1, b1, Wf, bf=None):
    """pack_made_backward's dict for the plain MLP; offsets: (W0, b0), (W1, b1), (Wf, bf)."""
    sl = _mlp_layers(W0, b0, W1, b1, Wf, bf)
    return None if sl is None else _pack_backward_from(sl)


# ---- the packs as GATHER INDICES (training: parameters change every step; the streams are rebuilt on the device) ------------------------
def index_arrays(shapes):
    """[(weight index array, bias index array)] for linears of the given (out, in) shapes over the flat parameter vector
    [0, W_0.flatten(), b_0, W_1.flatten(), b_1, ...] (element 0 = the zero every padded / masked stream entry points at)."""
    out, off = [], 1
    for o, i in shapes:
        w = off + np.arange(o * i, dtype=np.int64).reshape(o, i)
        off += o * i
        b = off + np.arange(o, dtype=np.int64)
        off += o
        out.append((w, b))
    assert off < 2 ** 24          # (the packers move float32 arrays: indices stay exact)
    return out


def _as_src(blob):
    src = np.rint(blob).astype(np.int32)
    assert np.array_equal(src.astype(np.float32), blob)
    return src


def train_structure(sl_values, sl_index, padded_rows=False):
    """The value-independent part of a network's training packs: tables + gather indices `src_fwd` / `src_bwd` (stream entry ->
    position in the flat parameter vector of index_arrays) such that flat[src] reproduces the packed streams of the value packers."""
    fwd_v = _pack_forward_from(sl_values)
    fwd_i = _pack_forward_from(sl_index)
    bwd = _pack_backward_from(sl_values)
    bwd_i = _pack_backward_from(sl_index)
    assert np.array_equal(fwd_v[1], fwd_i[1]) and np.array_equal(bwd["table"], bwd_i["table"])
    bwd["src"] = _as_src(bwd_i["blob"])
    if padded_rows:      # x, g_params and g_x live in rows padded to the weight-gradient tiles (hdr[14], hdr[15]: row strides)
        fwd_v[1][14] = bwd["Dx"]
        bwd["table"][14], bwd["table"][15] = bwd["Mp"], bwd["Dx"]
    return dict(table=fwd_v[1], src=_as_src(fwd_i[0]), hp=int(fwd_v[1][3]), bwd=bwd)


_STRUCTS = {}        # value-independent structures by what they depend on (masks / shapes): the layers of a model share them


def _shared(key, build):
    if key not in _STRUCTS:
        if len(_STRUCTS) > 64:
            _STRUCTS.clear()
        _STRUCTS[key] = build()
    st = _STRUCTS[key]
    if st is None:
        return None
    if isinstance(st, dict):          # the caller replaces entries by device tensors: hand out copies of the containers
        st = dict(st)
        st["bwd"] = dict(st["bwd"])
    return st


def _mask_key(made):
    import hashlib
    h = hashlib.sha1()
    for l in [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers] + [made.final_layer]:
        h.update(np.ascontiguousarray(l.mask.cpu().numpy() != 0).tobytes())
        h.update(str(tuple(l.mask.shape)).encode())
    return h.hexdigest()


def made_train_structure(made, mult):
    def build():
        sl = _slot_layers(made, mult)
        if sl is None:
            return None
        lins = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers] + [made.final_layer]
        return train_structure(sl, _slot_layers(made, mult, wb=index_arrays([tuple(l.weight.shape) for l in lins])))
    if not supported(made, mult):
        return None
    return _shared(("made", mult, _mask_key(made)), build)


def resnet_train_structure(net):
    def build():
        sl = _dense_layers(net)
        lins = [net.initial_layer] + [l for b in net.blocks for l in b.linear_layers] + [net.final_layer]
        return train_structure(sl, _dense_layers(net, wb=index_arrays([tuple(l.weight.shape) for l in lins])))
    if not resnet_supported(net):
        return None
    return _shared(("resnet", net.initial_layer.in_features, net.initial_layer.out_features, len(net.blocks),
                    net.final_layer.out_features), build)


def convnet_train_structure(cin, hid, cout):
    """3x3 -> 1x1 -> 3x3 conv conditioner over pixel rows (plain-MLP mode); flat vector = [0, conv1.weight, conv1.bias, conv2.weight,
    conv2.bias, conv3.weight] flattened in their own layouts."""
    return _shared(("conv", cin, hid, cout), lambda: _convnet_train_structure(cin, hid, cout))


def _convnet_train_structure(cin, hid, cout):
    o1 = 1
    w1 = (o1 + np.arange(hid * cin * 9, dtype=np.int64)).reshape(hid, cin, 3, 3).transpose(0, 2, 3, 1).reshape(hid, 9 * cin)
    ob1 = o1 + hid * cin * 9
    b1 = ob1 + np.arange(hid, dtype=np.int64)
    o2 = ob1 + hid
    w2 = (o2 + np.arange(hid * hid, dtype=np.int64)).reshape(hid, hid)
    ob2 = o2 + hid * hid
    b2 = ob2 + np.arange(hid, dtype=np.int64)
    o3 = ob2 + hid
    w3 = (o3 + np.arange(cout * hid * 9, dtype=np.int64)).reshape(cout, hid, 3, 3).transpose(2, 3, 0, 1).reshape(9 * cout, hid)
    assert o3 + cout * hid * 9 < 2 ** 24
    ones = lambda a: np.ones(a.shape, dtype=np.float32)
    sl_v = _mlp_layers(ones(w1), ones(b1), ones(w2), ones(b2), ones(w3), None)
    if sl_v is None:
        return None
    # weight gradients straight into the conv parameters' (o, c, ky, kx) layouts: destination = row position + column position
    tap, ch1, ch3 = np.arange(9), np.arange(cin), np.arange(hid)
    sl_v["wmaps"] = [
        (np.arange(hid) * cin * 9, (ch1[None, :] * 9 + tap[:, None]).reshape(-1), 1, np.arange(hid)),          # conv1: W1c[o][tap cin + c]
        (np.arange(hid) * hid, np.arange(hid), 1, np.arange(hid)),                                                # conv2: (o, c, 1, 1)
        ((np.arange(cout)[None, :] * hid * 9 + tap[:, None]).reshape(-1), ch3 * 9, 1, np.arange(9 * cout)),     # conv3: W3t[tap cout + o][c]
    ]
    return train_structure(sl_v, _mlp_layers(w1, b1, w2, b2, w3, None), padded_rows=True)


def maf_inverse_structure(made, blocks=(1, 2, 3), tri=False):
    """(gather indices, table) of the one-pass inverse kernel's pack (flows/maf_pack.pack_made: a pure rearrangement as well) over
    the same flat parameter vector as the training structures -- the packer run on a copy of the MADE that holds parameter
    positions instead of values; None outside that packer's structure.  `tri`: format 1 (regular tiles triangular)."""
    import copy
    from . import maf_pack
    if not maf_pack.supported(made, 2, blocks):
        return None
    key = ("maf_inverse", tuple(blocks), bool(tri), _mask_key(made))
    if key in _STRUCTS:
        return _STRUCTS[key]
    _STRUCTS[key] = _maf_inverse_structure(made, blocks, tri)
    return _STRUCTS[key]


def _maf_inverse_structure(made, blocks, tri=False):
    import copy
    from . import maf_pack
    if maf_pack.pack_made(made, blocks=blocks, tri=tri) is None:
        return None
    twin = copy.deepcopy(made).cpu()
    lins = [twin.initial_layer] + [l for b in twin.blocks for l in b.linear_layers] + [twin.final_layer]
    with torch.no_grad():
        for lin, (w, b) in zip(lins, index_arrays([tuple(l.weight.shape) for l in lins])):
            lin.weight.copy_(torch.from_numpy(w.astype(np.float32)))
            lin.bias.copy_(torch.from_numpy(b.astype(np.float32)))
    blob, table = maf_pack.pack_made(twin, blocks=blocks, tri=tri)
    return _as_src(blob), table


def maf_solve_t_structure(made, blocks=(1, 2, 3), tri=False):
    """(gather indices, table) of the transposed one-pass solve's pack (flows/maf_pack.pack_made_transposed) over the flat parameter
    vector of index_arrays -- the packer run on a copy of the MADE that holds parameter positions instead of values."""
    import copy
    from . import maf_pack
    if not maf_pack.supported(made, 2, blocks):
        return None
    key = ("maf_solve_t", tuple(blocks), bool(tri), _mask_key(made))
    if key in _STRUCTS:
        return _STRUCTS[key]
    st = None
    if maf_pack.pack_made(made, blocks=blocks) is not None:
        twin = copy.deepcopy(made).cpu()
        lins = [twin.initial_layer] + [l for b in twin.blocks for l in b.linear_layers] + [twin.final_layer]
        with torch.no_grad():
            for lin, (w, b) in zip(lins, index_arrays([tuple(l.weight.shape) for l in lins])):
                lin.weight.copy_(torch.from_numpy(w.astype(np.float32)))
                lin.bias.copy_(torch.from_numpy(b.astype(np.float32)))
        blob, table = maf_pack.pack_made_transposed(twin, blocks=blocks, tri=tri)
        st = (_as_src(blob), table)
    _STRUCTS[key] = st
    return st
