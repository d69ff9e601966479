"""Host-side packing for the one-launch NSF coupling layer on 64-row tiles (nf_nsf_wide, csrc/nsf_wide.hip): the shapes of
CoupledRationalQuadraticSpline (normflows/flows/neural_spline/wrapper.py:20-35 over nets/resnet.py:53-104) beyond the benchmark
kernel's (csrc/rqs_fused.hip: D <= 64, hidden <= 128) -- up to 128 features and 512 hidden units, 4 / 8 / 16 bins, linear tails.

This module only rearranges weights (no arithmetic on data besides the constant log2(e) / sqrt(hidden) folded into the width / height
rows, nsf/coupling.py:334-339).  Geometry (csrc/mlp_tile.hpp, nsf_wide.hip):
  * hidden units zero-padded to Hp = 128 | 256 | 512; row-block rb = units [32 rb, 32 rb + 32); k-group = 8 consecutive inputs; the
    k-loops of the hidden -> hidden and final products run over the hidden width rounded up to 32 (round 5), not over Hp;
  * the x tile is held with its columns SORTED: position i < PI = identity feature i, position PI + j = transform feature j
    (PI, PT = the two counts rounded up to 32; Dp = PI + PT; padding positions hold zeros).  The initial layer contracts over the
    first PI positions only -- the conditioner sees the identity features alone (nsf/coupling.py:83-84): a NaN in a TRANSFORM
    column must not reach it through a zero weight -- and the LU layer's dense matrix is packed in the same position order;
  * hidden work items per wave w: Hp 128 (128-row tiles): row-block w & 3 for the sample blocks 2 (w >> 2), 2 (w >> 2) + 1; Hp 256:
    row-block w, both sample blocks; Hp 512: row-blocks w and w + 8, both sample blocks;
  * the final layer is cut into GROUPS of 4 transform features = 3 row-blocks (96 MFMA rows, 92 used): accumulator register `reg` of
    row-block r3 in lane-half hh is slot v = 16 r3 + reg of the lane's parameter list, feature tf = 4 g + 2 hh + v // 24, parameter
    v % 24 (8 widths | 8 heights | 7 derivatives | pad) -- a lane ends up with the 2 x 24 parameters of two whole features, in the
    order the register spline routine (fused_common.hpp rqs_regs) reads them; group g belongs to wave g % 8 (both sample blocks);
  * an item's stream = its bias group(s) (4 KB per row-block) followed, per k-group, by one 1 KB A fragment per row-block; a wave's
    stream = its items in consumption order + a copy of its first 8 entries (the register ring wraps into the next tile).

  * optionally the adjacent LULinearPermute (mixing.py:535-563) as ONE dense D x D product on the tile (W, b composed in float64 by
    the layer: density y = L U x[perm] + b, sampling y = P^T U^-1 L^-1 (x - b)): an item of row-block w & 3 (32 output columns) for
    sample block w >> 2, FIRST in the stream in the density direction (core.py:193-195 visits the LU layer before its coupling
    layer), LAST in the sampling direction -- so a pack is per direction.

int32 table : hdr[32] = [D, Dp, H, Hp, NB, nI, nT, par_i, par_t, G, nfi, total floats, nhi, has_lu, TR, PI], hdr[16 + w] = offset (floats)
              of wave w's stream, hdr[24] = bins; then per wave: [LU entry (density)] | (1 + 2 NB) nhi hidden entries [nkg, rb, sb0] | nfi final
              entries [nkg, g, sb0] (g = -1: none; a group's item covers sample blocks sb0, sb0 + 1) | [LU entry (sampling)]; LU entry =
              [nkg, rb, sb0] (rb = -1: none).
"""
import numpy as np
import torch

from .made_pack import ROWS, KG, RING, a_stream, bias_group

HDR = 32
K_BINS = 8              # the default; 4 and 16 bins ride the same schedule (round 5): a lane's 48 accumulator values per sample block
M = 3 * K_BINS - 1      # are 48 / (3 K) whole parameter lists -- K = 4: four features of 12 slots, 8: two of 24, 16: one of 48 -- so a
MP = 3 * K_BINS         # group (3 row-blocks, both lane-halves) holds 8 / 4 / 2 transform features
SUPPORTED_BINS = (4, 8, 16)


def bins_geometry(K):
    """(parameters per feature, slots per feature, features per lane-half, features per group, final items a wave may own)."""
    mp = 3 * K
    fpl = 48 // mp
    return 3 * K - 1, mp, fpl, 2 * fpl, (8 if K == 16 else 4)


def geometry(Hp):
    """(hidden items per wave, sample blocks per item, rows per tile).  A 128-wide network's activations (64 KB for 128 rows) leave
    room for 128-row tiles: half the barriers, tile prologues and weight-stream traffic per row."""
    return {128: (1, 2, 128), 256: (1, 2, 64), 512: (2, 2, 64)}[Hp]


def hidden_item(Hp, w, i):
    """(row-block, first sample block) of hidden item i of wave w."""
    if Hp == 128:
        return w & 3, 2 * (w >> 2)
    if Hp == 256:
        return w, 0
    return w + 8 * i, 0


def final_row(g, r3, rho, nT, K=K_BINS):
    """Row of the ((3 K - 1) nT, hidden) final weight held by MFMA row rho of row-block r3 of group g, or -1 (padding)."""
    m, mp, fpl, fpg, _ = bins_geometry(K)
    q, hh, i = rho >> 3, (rho >> 2) & 1, rho & 3
    v = 16 * r3 + 4 * q + i
    f, prm = v // mp, v % mp
    tf = fpg * g + fpl * hh + f
    if prm >= m or tf >= nT:
        return -1
    return tf * m + prm


def supported(prqct):
    from .. import nets
    net = prqct.transform_net
    if not (isinstance(net, nets.ResidualNet) and net.is_plain_relu()):
        return False
    if prqct.tails != "linear" or getattr(prqct, "_per_feature", False) or prqct.unconditional_transform is None:
        return False
    K = prqct.num_bins
    if K not in SUPPORTED_BINS or prqct.min_bin_width * K > 1.0 or prqct.min_bin_height * K > 1.0:
        return False
    D = prqct.features
    if not (2 <= D <= 128 and 1 <= net.hidden_features <= 512 and len(net.blocks) >= 1 and len(net.blocks) <= 7):
        return False
    if net.initial_layer.weight.dtype != torch.float32:
        return False
    ii, ti = prqct.identity_features.cpu(), prqct.transform_features.cpu()
    alt0 = torch.equal(ii, torch.arange(0, D, 2)) and torch.equal(ti, torch.arange(1, D, 2))
    alt1 = torch.equal(ii, torch.arange(1, D, 2)) and torch.equal(ti, torch.arange(0, D, 2))
    return alt0 or alt1


def pack_nsf_wide(prqct, lu=None, direction=0):
    """(blob float32 ndarray, table int32 ndarray) or None (the caller keeps the layer-wise path).  lu = (W (D, D), b (D,)) numpy
    arrays of the adjacent LULinearPermute in `direction` (0 density, 1 sampling), or None."""
    if not supported(prqct):
        return None
    net = prqct.transform_net
    D, H, NB = prqct.features, net.hidden_features, len(net.blocks)
    ident = prqct.identity_features.cpu().numpy()
    trans = prqct.transform_features.cpu().numpy()
    nI, nT = len(ident), len(trans)
    par_i, par_t = int(ident[0]), int(trans[0])
    Hp = 128 if H <= 128 else (256 if H <= 256 else 512)
    PI, PT = (nI + 31) // 32 * 32, (nT + 31) // 32 * 32
    Dp = PI + PT
    col_of = -np.ones(Dp, dtype=np.int64)          # position -> column of the row (or -1: padding)
    col_of[:nI] = ident
    col_of[PI:PI + nT] = trans
    nhi, NS, TR = geometry(Hp)
    K = prqct.num_bins
    M_, MP_, FPL, FPG, nfi_max = bins_geometry(K)
    G = (nT + FPG - 1) // FPG
    nsp = TR // 64                      # pairs of sample blocks per tile: a final item = (group, pair)
    nfi = (G * nsp + 7) // 8
    if nfi > nfi_max:                   # (the kernel keeps a wave's log-det sums per final item in registers)
        return None
    f32 = lambda t: t.detach().cpu().numpy().astype(np.float32)

    W0 = np.zeros((Hp, PI), dtype=np.float32)
    W0[:H, :nI] = f32(net.initial_layer.weight)
    b0 = np.zeros(Hp, dtype=np.float32)
    b0[:H] = f32(net.initial_layer.bias)
    # round 5: the CONTRACTION extent of the hidden -> hidden and final products is the hidden width rounded up to 32 (a k-loop runs
    # in steps of four k-groups), not Hp: a 192-wide network keeps Hp = 256 for its row-blocks (8 waves x 32 units) but its k-loops
    # run over 24 k-groups instead of 32 -- the padding is no longer paid for in K (D 96 / hidden 192: 675 -> see profiles/r05_*)
    Kh = (H + 31) // 32 * 32
    layers = [(W0, b0)]
    for blk in net.blocks:
        for lin in blk.linear_layers:
            W = np.zeros((Hp, Kh), dtype=np.float32)
            W[:H, :H] = f32(lin.weight)
            b = np.zeros(Hp, dtype=np.float32)
            b[:H] = f32(lin.bias)
            layers.append((W, b))
    wf, bf = f32(net.final_layer.weight), f32(net.final_layer.bias)        # (23 nT, H)
    if wf.shape[0] != M_ * nT:
        return None
    wh_scale = np.float32(1.4426950408889634 / np.sqrt(float(H)))          # log2(e) / sqrt(hidden): rqs_regs takes exp2
    WF = np.zeros((G, 3, ROWS, Kh), dtype=np.float32)
    BF = np.zeros((G, 3, ROWS), dtype=np.float32)
    for g in range(G):
        for r3 in range(3):
            for rho in range(ROWS):
                row = final_row(g, r3, rho, nT, K)
                if row >= 0:
                    sc = wh_scale if (row % M_) < 2 * K else np.float32(1.0)
                    WF[g, r3, rho, :H] = wf[row] * sc
                    BF[g, r3, rho] = bf[row] * sc

    nhl = 1 + 2 * NB
    has_lu = lu is not None
    nitems = nhl * nhi + nfi + (1 if has_lu else 0)
    base = 1 if (has_lu and direction == 0) else 0
    if has_lu:
        valid = col_of >= 0
        WL = np.zeros((Dp, Dp), dtype=np.float32)
        WL[np.ix_(valid, valid)] = np.asarray(lu[0], dtype=np.float32)[np.ix_(col_of[valid], col_of[valid])]
        bL = np.zeros(Dp, dtype=np.float32)
        bL[valid] = np.asarray(lu[1], dtype=np.float32)[col_of[valid]]
    hdr = np.zeros(HDR, dtype=np.int32)
    tab = np.zeros((8, nitems, 3), dtype=np.int32)
    chunks, off = [], 0

    def lu_item(w, stream, idx):
        rb, sb0 = w & 3, (w >> 2) * (TR // 64)          # TR = 128: both sample blocks of the wave's half of the tile
        if rb >= Dp // ROWS:
            tab[w, idx] = (0, -1, 0)
            return
        tab[w, idx] = (Dp // KG, rb, sb0)
        stream.append(bias_group(bL[rb * ROWS:(rb + 1) * ROWS]))
        stream.append(a_stream(WL[rb * ROWS:(rb + 1) * ROWS]))

    for w in range(8):
        hdr[16 + w] = off
        stream = []
        if has_lu and direction == 0:
            lu_item(w, stream, 0)
        for l in range(nhl):
            Wl, bl = layers[l]
            nkg = Wl.shape[1] // KG
            for i in range(nhi):
                rb, sb0 = hidden_item(Hp, w, i)
                tab[w, base + l * nhi + i] = (nkg, rb, sb0)
                stream.append(bias_group(bl[rb * ROWS:(rb + 1) * ROWS]))
                stream.append(a_stream(Wl[rb * ROWS:(rb + 1) * ROWS]))
        for j in range(nfi):
            g, sp = divmod(w + 8 * j, nsp)
            if g >= G:
                tab[w, base + nhl * nhi + j] = (0, -1, 0)
                continue
            nkg = Kh // KG
            tab[w, base + nhl * nhi + j] = (nkg, g, 2 * sp)
            for r3 in range(3):
                stream.append(bias_group(BF[g, r3]))
            frag = np.stack([a_stream(WF[g, r3]).reshape(nkg, 256) for r3 in range(3)], axis=1)   # [nkg][3][256]
            stream.append(frag.reshape(-1))
        if has_lu and direction == 1:
            lu_item(w, stream, nitems - 1)
        stream = np.concatenate(stream)
        stream = np.concatenate([stream, np.resize(stream, RING * 256)])
        chunks.append(stream)
        off += stream.size
    hdr[:16] = [D, Dp, H, Hp, NB, nI, nT, par_i, par_t, G, nfi, off, nhi, int(has_lu), TR, PI]
    hdr[24] = K
    blob = np.concatenate(chunks).astype(np.float32)
    assert blob.size == off and off < 2 ** 31
    return blob, np.concatenate([hdr, tab.reshape(-1)]).astype(np.int32)
