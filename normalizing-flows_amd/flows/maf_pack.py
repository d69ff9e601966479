"""Host-side packing for the incremental MAF inverse kernel (nf_maf_inverse, csrc/maf_inverse.hip).

The reference inverts a masked autoregressive layer with D full MADE passes (normflows/flows/affine/
autoregressive.py:29-38).  Because hidden unit j of degree m only sees inputs of degree <= m (nets/made.py:63-81),
every unit can be finalised ONCE, right after feature m is known: total work = one MADE pass.  The kernel walks the
degrees in order; hidden units are sorted by degree and cut into tiles of <= 32 units that hold whole degrees:

  * contributions from earlier tiles are dense 32-row GEMM blocks (MFMA; "block part", weights stored in the
    MFMA A-operand order [k/8][half][unit][4]),
  * contributions inside the tile (the 32x32 diagonal blocks, masked) run in the sequential per-degree part
    (weights read as scalars).

This module only rearranges weights (no arithmetic on data).  Layout of the float blob (all offsets in floats):

  [0:4]                       final-layer bias of feature 0 (unconstrained scale, shift) + 2 pad: depends on no hidden unit
  per tile t, at table[t].rec : A0 [K0/8][2][32][4] | A1..A4, AF [4t][2][32][4] each | bias[5][32] | biasF[32]
                                | W0d[32][16] | Wd[4][32][32] | WFd[32][32]
  (num_blocks = 2: five hidden layers.  In general NL = 1 + 2 num_blocks hidden layers: A1..A_{NL-1}, AF | bias[NL][32] | biasF
  | W0d | Wd[NL-1][32][32] | WFd; num_blocks in 1..3 is taken by nf_maf_inverse_h, round 2's nf_maf_inverse / the rows layout
  only num_blocks = 2)
and of the int32 table: [D, Dp, H, Hp, T, mult, num_blocks, 0] then per tile 24 ints
  [dlo, nsteps, K0, rec, mask[0..15], 0, 0, 0, 0]   (mask[s] = bitmask of the tile's units that have degree dlo+s).

With `mult` = 2 (affine: unconstrained scale, shift) the final-layer rows of a tile's <= 16 features fit ONE 32-row
block (layout above).  With `mult` > 2 (autoregressive spline: 3K-1 | 3K | 3K+1 <= 32 numbers per feature, kernel
nf_arnsf_inverse) every feature gets its own 32-row block, zero-padded:

  [0:32]                      final-layer bias of feature 0
  per tile t                  A0 | A1..A4 [4t][2][32][4] | AF_s [4t][2][32][4] for each step s | bias[5][32]
                              | W0d[32][16] | Wd[4][32][32] | biasF[nsteps][32] | WFd[nsteps][mult][32]

Format 1 (`tri=True`, table[7] = 1; nf_maf_inverse_h only, round 5): REGULAR tiles -- at most FAST_STEPS = 8 degrees of at most 4
units each (config 5: 15 of the 16 tiles) -- carry their sequential part's weights TRIANGULAR and in the order the kernel's
statically unrolled steps read them; table entry [20] = 1 marks such a tile.  Unit i of step g sits at the MFMA row whose
accumulator is register 2 g + (i & 1) of lane-half i >> 1 (row = (r & 3) + 8 (r >> 2) + 4 half), so after step g registers
0 .. 2 g + 1 of BOTH halves are final and a step's four targets are two per half.  Record of a regular tile after the A operands:
  bias[NL][32] | biasF[32] (row g = unconstrained scale of step g's feature on half 0, row g + 4.. = its shift: same register, half 1)
  | half 0 | half 1, each half = W0f | Wdf[NL-1] | WFf in float4 units:
     W0f  step g, q = 0..g/2      : (w[t0][x 2q], w[t0][x 2q+1], w[t1][x 2q], w[t1][x 2q+1])   t0, t1 = the half's OWN targets, x = window
     Wdf  step g, q = 0..g, j=0,1 : (w[t 2j][s 2q], w[t 2j][s 2q+1], w[t 2j+1][s 2q], w[t 2j+1][s 2q+1])   t = all four targets,
                                     s 2q, s 2q+1 = the half's registers 2q, 2q+1 of the source layer (slot q <= g only: triangular)
     WFf  step g, q = 0..g        : (w[us][s 2q], w[us][s 2q+1], w[sh][s 2q], w[sh][s 2q+1])
Other tiles keep the format-0 record.
"""
import numpy as np
import torch
from torch.nn import functional as F

TILE = 32        # hidden units per tile (MFMA rows)
MAX_STEPS = 16   # degrees per tile (two final-layer rows per step -> 32 MFMA rows)
TABLE_HDR = 8
TABLE_ENT = 24
FAST_STEPS = 8   # format 1: a REGULAR tile has <= 8 degrees of <= 4 units
FAST_W0 = sum(g // 2 + 1 for g in range(FAST_STEPS))          # float4 per half
FAST_WD = sum(2 * (g + 1) for g in range(FAST_STEPS))
FAST_WF = sum(g + 1 for g in range(FAST_STEPS))


def fast_half_floats(NL):
    return 4 * (FAST_W0 + (NL - 1) * FAST_WD + FAST_WF)


def _row_of(reg, half):
    """MFMA row whose accumulator is register `reg` of lane-half `half` (v_mfma_f32_32x32x2_f32 C layout)."""
    return (reg & 3) + 8 * (reg >> 2) + 4 * half


def is_regular(steps):
    return len(steps) <= FAST_STEPS and max(steps) <= 4


# Format 1, tile kind 2 ("regular with extras", round 5): at most 7 degrees; the first m <= 4 of them own FIVE units, the others at
# most four.  The four regular units of a step sit where a regular tile has them; the fifth unit of step g < m sits in the otherwise
# empty slot 7: lane-half g & 1, accumulator register 14 + (g >> 1).  BASELINE configs[4]'s tile 0 (degrees 1-4 own five units) is
# such a tile: with it every tile of that layer runs the statically unrolled sequential part and the layer is ONE launch.
XTRA_STEPS = 7
XTRA_MAX = 4


def extras_prefix(steps):
    """m if `steps` is a kind-2 tile (m >= 1 leading five-unit degrees), else 0."""
    if len(steps) > XTRA_STEPS or max(steps) > 5 or max(steps) < 5:
        return 0
    m = 0
    while m < len(steps) and steps[m] == 5:
        m += 1
    if m > XTRA_MAX or any(c > 4 for c in steps[m:]):
        return 0
    return m


def xtra_half4(NL):
    """float4 units of the extra region of one half: X1 (pair 7 -> the four regular targets) [NL-1][7][2] | X2 (the extra target over
    pairs 0..g and 7) [NL-1][4][3] | XW (the extra target's window) [4] | XF (pair 7 -> scale / shift rows) [7]."""
    return (NL - 1) * (XTRA_STEPS * 2 + XTRA_MAX * 3) + XTRA_MAX + XTRA_STEPS


def tile_row(g, i):
    """Row (within the tile) of unit i of step g in a regular (i < 4) or kind-2 (i == 4: the extra) tile."""
    if i < 4:
        return _row_of(2 * g + (i & 1), i >> 1)
    return _row_of(14 + (g >> 1), g & 1)


def plan_tiles(D, hidden_degrees):
    """Sort units by degree (stable) and cut into tiles.  Returns (order, tiles) with tiles = list of
    (dlo, nsteps, [unit-count per step]) or None when the structure is outside what the kernel handles."""
    deg = np.asarray(hidden_degrees, dtype=np.int64)
    if D < 2 or deg.min() < 1 or deg.max() > D - 1:
        return None
    counts = np.bincount(deg, minlength=D)[1:D]          # degrees 1..D-1
    if (counts < 1).any() or (counts > TILE).any():
        return None                                       # every degree needs 1..32 units
    order = np.argsort(deg, kind="stable")
    tiles, d = [], 1
    while d <= D - 1:
        n_units, steps = 0, []
        while d <= D - 1 and len(steps) < MAX_STEPS and n_units + counts[d - 1] <= TILE:
            steps.append(int(counts[d - 1]))
            n_units += counts[d - 1]
            d += 1
        tiles.append((d - len(steps), len(steps), steps))
    return order, tiles


def _a_operand(w_rows_by_k):
    """(32, K) row-major block -> MFMA A-operand order [K/8][2][32][4] (k = 8*kb + 4*half + i)."""
    rows, K = w_rows_by_k.shape
    assert rows == TILE and K % 8 == 0
    return np.ascontiguousarray(w_rows_by_k.reshape(TILE, K // 8, 2, 4).transpose(1, 2, 0, 3)).reshape(-1)


def supported(made, mult=2, blocks=(2,)):
    from .. import nets
    if not isinstance(made, nets.MADE):
        return False
    if not isinstance(made.preprocessing, torch.nn.Identity) or hasattr(made, "context_layer"):
        return False
    if len(made.blocks) not in blocks or not all(isinstance(b, nets.MaskedResidualBlock) for b in made.blocks):
        return False
    for b in made.blocks:
        if b.use_batch_norm or b.activation is not F.relu or b.dropout.p != 0.0 or hasattr(b, "context_layer"):
            return False
    D = made.initial_layer.in_features
    if made.final_layer.out_features != mult * D or made.initial_layer.weight.dtype != torch.float32:
        return False
    return 2 <= mult <= TILE


def pack_made(made, mult=2, rows=False, blocks=(2,), tri=False):
    """Returns (blob float32 ndarray, table int32 ndarray) or None if the MADE is not the supported structure.
    `mult` = final-layer outputs per feature (MADE's output_multiplier); `rows` selects the one-block-per-feature
    layout of nf_arnsf_inverse; `blocks`: the residual-block counts the caller's kernel takes; `tri`: format 1 (module docstring;
    nf_maf_inverse_h only)."""
    if not supported(made, mult, blocks) or (rows and len(made.blocks) != 2):
        return None
    if tri and (rows or mult != 2):
        return None
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    hid_deg = made.initial_layer.degrees.cpu().numpy()
    lin = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers]
    NB = len(made.blocks)
    for l in lin[1:]:
        if not np.array_equal(l.degrees.cpu().numpy(), hid_deg):
            return None
    fin = made.final_layer
    # input degrees must be arange(1..D) and the output rows feature-major (`mult` rows per feature)
    m0 = made.initial_layer.mask.cpu().numpy()
    if not np.array_equal(m0, (hid_deg[:, None] >= np.arange(1, D + 1)[None, :]).astype(m0.dtype)):
        return None
    mf = fin.mask.cpu().numpy()
    out_deg = np.repeat(np.arange(1, D + 1), mult)
    if not np.array_equal(mf, (out_deg[:, None] > hid_deg[None, :]).astype(mf.dtype)):
        return None
    plan = plan_tiles(D, hid_deg)
    if plan is None:
        return None
    order, tiles = plan
    T = len(tiles)
    Hp, Dp = T * TILE, (D + 31) // 32 * 32

    # position of every sorted unit in the padded index space, and the degree of every padded slot (0 = pad)
    pos = np.zeros(H, dtype=np.int64)
    slot_deg = np.zeros(Hp, dtype=np.int64)
    k = 0
    kind = [0 if not tri else (1 if is_regular(steps) else (2 if extras_prefix(steps) else 0)) for (_, _, steps) in tiles]
    regular = [k != 0 for k in kind]          # tiles on the statically unrolled sequential part
    for t, (dlo, ns, steps) in enumerate(tiles):
        base = t * TILE
        for s, c in enumerate(steps):
            for i in range(c):
                if regular[t]:
                    slot = t * TILE + tile_row(s, i)
                else:
                    slot = base
                pos[k] = slot
                slot_deg[slot] = dlo + s
                base += 1
                k += 1
    assert k == H
    unit_of_slot = -np.ones(Hp, dtype=np.int64)
    unit_of_slot[pos] = order

    def padded(lyr, in_map, in_size):
        """Masked weight and bias of `lyr` with rows moved to padded slots and columns to `in_map` positions."""
        w = (lyr.weight.detach() * lyr.mask).cpu().numpy().astype(np.float32)
        b = lyr.bias.detach().cpu().numpy().astype(np.float32)
        W = np.zeros((Hp, in_size), dtype=np.float32)
        Bv = np.zeros(Hp, dtype=np.float32)
        rows = unit_of_slot >= 0
        tmp = np.zeros((int(rows.sum()), in_size), dtype=np.float32)
        tmp[:, in_map] = w[unit_of_slot[rows]]
        W[rows] = tmp
        Bv[rows] = b[unit_of_slot[rows]]
        return W, Bv

    feat_map = np.arange(D)
    hid_map = np.zeros(H, dtype=np.int64)
    hid_map[order] = pos            # original hidden index -> padded slot
    W0, b0 = padded(lin[0], feat_map, Dp)
    Wh, bh = [], []
    for l in lin[1:]:
        w, b = padded(l, hid_map, Hp)
        Wh.append(w)
        bh.append(b)
    wf = (fin.weight.detach() * fin.mask).cpu().numpy().astype(np.float32)   # (mult D, H), row mult f + p
    bf = fin.bias.detach().cpu().numpy().astype(np.float32)
    WF = np.zeros((mult * D, Hp), dtype=np.float32)
    WF[:, hid_map] = wf

    if mult != 2 and not rows:
        return None                        # the per-tile final block holds exactly two rows per feature
    head = np.zeros(TILE if rows else 4, dtype=np.float32)   # 16-byte multiples keep every section aligned
    head[:mult] = bf[:mult]
    chunks = [head]
    off = head.size
    table = np.zeros(TABLE_HDR + TABLE_ENT * T, dtype=np.int32)
    table[0:8] = [D, Dp, H, Hp, T, mult, NB, int(bool(tri))]
    for t, (dlo, ns, steps) in enumerate(tiles):
        r0, r1 = t * TILE, (t + 1) * TILE
        nprev = dlo - 1                    # features (0-based) 0..dlo-2 come from the block part; dlo-1.. from the window
        K0 = (nprev + 31) // 32 * 32          # the kernel's k loop is unrolled by 32
        a0 = np.zeros((TILE, K0), dtype=np.float32)
        a0[:, :nprev] = W0[r0:r1, :nprev]
        rec = [_a_operand(a0)] if K0 else []
        for w in Wh:
            if t:
                rec.append(_a_operand(w[r0:r1, :r0]))
        # final rows of the tile's output features dlo..dlo+ns-1 (0-based)
        if rows:
            fo = np.zeros((ns, TILE, Hp), dtype=np.float32)
            bfo = np.zeros((ns, TILE), dtype=np.float32)
            for j in range(ns):
                f = dlo + j
                fo[j, :mult] = WF[mult * f:mult * (f + 1)]
                bfo[j, :mult] = bf[mult * f:mult * (f + 1)]
                if t:
                    rec.append(_a_operand(fo[j][:, :r0]))
        else:
            fo = np.zeros((TILE, Hp), dtype=np.float32)
            bfo = np.zeros(TILE, dtype=np.float32)
            for j in range(ns):
                f = dlo + j
                # format 0: rows 2 j, 2 j + 1; regular tile: register j of half 0 (scale) / half 1 (shift)
                ra, rb = (_row_of(j, 0), _row_of(j, 1)) if regular[t] else (2 * j, 2 * j + 1)
                fo[ra], fo[rb] = WF[2 * f], WF[2 * f + 1]
                bfo[ra], bfo[rb] = bf[2 * f], bf[2 * f + 1]
            if t:
                rec.append(_a_operand(fo[:, :r0]))
        rec.append(np.concatenate([b0[r0:r1]] + [b[r0:r1] for b in bh]))
        if not rows:
            rec.append(bfo)
        w0d = np.zeros((TILE, MAX_STEPS), dtype=np.float32)  # window features dlo-1 .. dlo-1+ns-1 (0-based)
        nwin = min(MAX_STEPS, D - (dlo - 1))
        w0d[:, :nwin] = W0[r0:r1, dlo - 1:dlo - 1 + nwin]
        # a unit of degree dlo+s must not see window features above its own degree: the mask already guarantees it
        if regular[t]:
            halves = []
            for hh in (0, 1):
                part = []
                trow = lambda g, i: _row_of(2 * g + (i & 1), i >> 1)          # row (within the tile) of unit i of step g
                for g in range(FAST_STEPS):                                      # W0f: the half's own targets over the window pairs
                    t0, t1 = trow(g, 2 * hh), trow(g, 2 * hh + 1)
                    for q in range(g // 2 + 1):
                        part.append([w0d[t0, 2 * q], w0d[t0, 2 * q + 1], w0d[t1, 2 * q], w0d[t1, 2 * q + 1]])
                for w in Wh:                                                     # Wdf: all four targets over the half's registers <= 2 g + 1
                    d = w[r0:r1, r0:r1]
                    for g in range(FAST_STEPS):
                        for q in range(g + 1):
                            sa, sb = _row_of(2 * q, hh), _row_of(2 * q + 1, hh)
                            for j in (0, 1):
                                ta, tb = trow(g, 2 * j), trow(g, 2 * j + 1)
                                part.append([d[ta, sa], d[ta, sb], d[tb, sa], d[tb, sb]])
                d = fo[:, r0:r1]
                for g in range(FAST_STEPS):                                      # WFf: (scale, shift) of step g's feature
                    ru, rs = _row_of(g, 0), _row_of(g, 1)
                    for q in range(g + 1):
                        sa, sb = _row_of(2 * q, hh), _row_of(2 * q + 1, hh)
                        part.append([d[ru, sa], d[ru, sb], d[rs, sa], d[rs, sb]])
                part = np.asarray(part, dtype=np.float32).reshape(-1)
                assert part.size == fast_half_floats(len(Wh) + 1)
                halves.append(part)
            rec.append(np.concatenate(halves))
            if kind[t] == 2:
                m = extras_prefix(steps)
                for hh in (0, 1):
                    part = []
                    r14, r15 = _row_of(14, hh), _row_of(15, hh)
                    for w in Wh:                           # X1: pair 7 of this half -> the four regular targets of step g
                        d = w[r0:r1, r0:r1]
                        for g in range(XTRA_STEPS):
                            for j in (0, 1):
                                ta, tb = trow(g, 2 * j), trow(g, 2 * j + 1)
                                part.append([d[ta, r14], d[ta, r15], d[tb, r14], d[tb, r15]])
                    for w in Wh:                           # X2: the extra target of step g over this half's pairs 0..g and 7
                        d = w[r0:r1, r0:r1]
                        for g in range(XTRA_MAX):
                            te_ = tile_row(g, 4)
                            pairs = [(_row_of(2 * q, hh), _row_of(2 * q + 1, hh)) for q in range(g + 1)] + [(r14, r15)]
                            flat = [d[te_, c] if g < m else 0.0 for pr in pairs for c in pr]
                            flat += [0.0] * (12 - len(flat))
                            for k in range(3):
                                part.append(flat[4 * k:4 * k + 4])
                    for g in range(XTRA_MAX):              # XW: the extra target's window weights (pairs 0 .. g / 2), own half only
                        te_ = tile_row(g, 4)
                        own = g < m and hh == (g & 1)
                        flat = [w0d[te_, c] if own and c <= g + 1 else 0.0 for c in range(4)]
                        part.append(flat)
                    d = fo[:, r0:r1]
                    for g in range(XTRA_STEPS):            # XF: pair 7 -> (scale, shift) of step g's feature
                        ru, rs = _row_of(g, 0), _row_of(g, 1)
                        part.append([d[ru, r14], d[ru, r15], d[rs, r14], d[rs, r15]])
                    part = np.asarray(part, dtype=np.float32).reshape(-1)
                    assert part.size == 4 * xtra_half4(len(Wh) + 1)
                    rec.append(part)
        else:
            rec.append(w0d.reshape(-1))
            for w in Wh:
                rec.append(np.ascontiguousarray(w[r0:r1, r0:r1]).reshape(-1))
        if regular[t]:
            pass
        elif rows:
            rec.append(bfo.reshape(-1))
            rec.append(np.ascontiguousarray(fo[:, :mult, r0:r1]).reshape(-1))
        else:
            rec.append(np.ascontiguousarray(fo[:, r0:r1]).reshape(-1))
        rec = np.concatenate(rec)
        e = TABLE_HDR + TABLE_ENT * t
        table[e + 0], table[e + 1], table[e + 2], table[e + 3] = dlo, ns, K0, off
        table[e + 20] = kind[t]
        table[e + 21] = extras_prefix(steps) if kind[t] == 2 else 0
        u = 0
        for s, c in enumerate(steps):
            table[e + 4 + s] = np.array([((1 << c) - 1) << u], dtype=np.uint64).astype(np.uint32).view(np.int32)[0]
            u += c
        chunks.append(rec)
        off += rec.size
    blob = np.concatenate(chunks).astype(np.float32)
    return blob, table


# ---- format 2 (round 5): the TRANSPOSED one-pass solve of the implicit backward (csrc/maf_solve_t.hip, autograd.MafInverseFn) ----
def _unit_positions(made, tri):
    """(fslot, vslot, T): per hidden unit its PADDED position in the inverse kernel's scratch (pack_made, format 0 / 1 with `tri`) and
    in the transposed solve's (pack_made_transposed: forward tiles reversed, forward positions within a tile); T tiles."""
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    hid_deg = made.initial_layer.degrees.cpu().numpy()
    order, tiles = plan_tiles(D, hid_deg)
    T = len(tiles)
    fslot = np.zeros(H, dtype=np.int64)
    k = 0
    for t, (dlo, ns, steps) in enumerate(tiles):
        b_ = 0
        perm = bool(tri) and (is_regular(steps) or extras_prefix(steps) > 0)
        for g, c in enumerate(steps):
            for i in range(c):
                fslot[order[k]] = t * TILE + (tile_row(g, i) if perm else b_)
                b_ += 1
                k += 1
    vslot = (T - 1 - fslot // TILE) * TILE + fslot % TILE
    return fslot, vslot, T


def solve_t_gradient_columns(made, blocks=(1, 2, 3), tri=False, forward=False):
    """int32 (Hp_train,): for column c of the training kernels' hidden tensors (flows/made_pack: units sorted by degree, stable) the
    PADDED position of that unit in the transposed solve's activation scratch (pack_made_transposed: forward tiles reversed, forward
    positions), -1 for the padding columns.  With it the solve's scratch IS every hidden layer's output gradient of MADE's
    input-gradient chain (nf_maf_scratch_rows rearranges it into G[l][rows][Hp_train]): the solve finalises every virtual unit once from
    final values, which is that chain at the solution.
    forward=True: the positions in the INVERSE kernel's scratch (pack_made, format 0 / 1 with `tri`): its published activations are
    relu(h_0), relu(t_0), relu(h_1), ..., h_NB -- the inputs of MADE's linears, what the weight-gradient launch multiplies G with."""
    if not supported(made, 2, blocks) or pack_made(made, blocks=blocks) is None:
        return None
    H = made.initial_layer.out_features
    hid_deg = made.initial_layer.degrees.cpu().numpy()
    fslot, vslot, _ = _unit_positions(made, tri)
    hp_train = 256 if H <= 256 else 512
    cols = np.full(hp_train, -1, dtype=np.int32)
    cols[:H] = (fslot if forward else vslot)[np.argsort(hid_deg, kind="stable")]
    return cols


def position_wgrad_tables(made, blocks=(1, 2, 3), tri=False):
    """Round 6: (wtable, stable, ntiles, positions) for nf_made_wgrad_pos -- the weight-gradient launch of the implicit backward
    reading the two one-pass kernels' scratches IN PLACE (csrc/made_bwd.hip made_wgrad_pos_kernel).  The problems are those of
    flows/made_pack._pack_backward_from in the same order (so its flat gradient layout, `offsets` and byte mask stay valid), but a
    hidden operand's slots are scratch POSITIONS: rows of dW_l (layer l's output gradient) = positions in the transposed solve's scratch,
    layer NL - 1 - l, negated; columns (the linear's input) = positions in the inverse pass's scratch, layer l - 1.  The row / column
    maps send a position to the parameter's own row / column (-1: a hole of a tile), the 128 x 128 tiles listed are those holding a
    mask non-zero in position space.  None outside the one-pass kernels' structures or when the position count is not a multiple of
    128 (the caller then rearranges: nf_maf_scratch_rows)."""
    if not supported(made, 2, blocks) or pack_made(made, blocks=blocks) is None:
        return None
    W_TILE = 128
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    NB = len(made.blocks)
    NL = 1 + 2 * NB
    MD = made.final_layer.out_features
    fslot, vslot, T = _unit_positions(made, tri)
    P = T * TILE
    if P % W_TILE:
        return None
    funit = np.full(P, -1, dtype=np.int32)
    vunit = np.full(P, -1, dtype=np.int32)
    funit[fslot] = np.arange(H)
    vunit[vslot] = np.arange(H)
    Mp = (MD + W_TILE - 1) // W_TILE * W_TILE
    Dx = (D + W_TILE - 1) // W_TILE * W_TILE
    feat_map = np.full(Dx, -1, dtype=np.int32)
    feat_map[:D] = np.arange(D)
    out_map = np.full(Mp, -1, dtype=np.int32)
    out_map[:MD] = np.arange(MD)
    lins = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers] + [made.final_layer]
    masks = [l.mask.cpu().numpy() != 0 for l in lins]
    SY, SX, NEG = 1, 2, 4
    # (dY base, dY matrix / scratch layer, ldY, X base, X matrix / scratch layer, ldX, relu, flags, mask, row map, column map, shape)
    probs = [(2, NL - 1, P, 1, 0, Dx, 0, SY | NEG, masks[0], vunit, feat_map, (H, D))]
    for b in range(NB):
        probs.append((2, NL - 1 - (2 * b + 1), P, 3, 2 * b, P, 1, SY | SX | NEG, masks[1 + 2 * b], vunit, funit, (H, H)))
        probs.append((2, NL - 1 - (2 * b + 2), P, 3, 2 * b + 1, P, 1, SY | SX | NEG, masks[2 + 2 * b], vunit, funit, (H, H)))
    probs.append((0, 0, Mp, 3, 2 * NB, P, 0, SX, masks[-1], out_map, funit, (MD, H)))
    ptab, tiles, stab, maps = [], [], [], []
    flat, map_off = 0, 8 * len(probs)
    for pi, (dyb, dyl, ldy, xb, xl, ldx, relu, flags, M, rmap, cmap, shape) in enumerate(probs):
        ptab.append([dyb, dyl, ldy, xb, xl, ldx, relu, flags])
        Mrows, Ncols = len(rmap), len(cmap)
        Mfull = np.zeros((Mrows, Ncols), dtype=bool)
        rr, cc = np.nonzero(rmap >= 0)[0], np.nonzero(cmap >= 0)[0]
        Mfull[np.ix_(rr, cc)] = M[np.ix_(rmap[rr], cmap[cc])]
        for mt in range(Mrows // W_TILE):
            nz = [nt for nt in range(Ncols // W_TILE) if Mfull[mt * W_TILE:(mt + 1) * W_TILE, nt * W_TILE:(nt + 1) * W_TILE].any()]
            if not nz:
                nz = [0]                                  # (the bias gradient still needs the tile row)
            for k, nt in enumerate(nz):
                tiles.append([pi, mt * W_TILE, nt * W_TILE, 1 if k == 0 else 0, 0, 0, 0, 0])
        woff, boff = flat, flat + shape[0] * shape[1]
        flat = boff + shape[0]
        stab.append([woff, shape[1], boff, map_off, map_off + Mrows, 0, 0, 0])
        maps += [np.asarray(rmap, dtype=np.int32), np.asarray(cmap, dtype=np.int32)]
        map_off += Mrows + Ncols
    whdr = np.zeros(16, dtype=np.int32)
    whdr[:2] = [len(tiles), len(probs)]
    wtable = np.concatenate([whdr, np.asarray(ptab, dtype=np.int32).reshape(-1), np.asarray(tiles, dtype=np.int32).reshape(-1)])
    stable = np.concatenate([np.asarray(stab, dtype=np.int32).reshape(-1)] + maps).astype(np.int32)
    return dict(wtable=wtable.astype(np.int32), stable=stable, ntiles=len(tiles), positions=P, nflat=flat, NL=NL)


def final_layer_columns(made):
    """int32 (2 D, Hp_train) gather indices into the flat parameter vector of flows/made_pack.index_arrays (0 = the zero entry): the
    MASKED final-layer weight with its columns in the training kernels' column order -- params = F.linear(h_NB, this, bias) reproduces
    MADE's output from the last hidden tensor (nets/made.py:296-304)."""
    from . import made_pack
    lins = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers] + [made.final_layer]
    wi = made_pack.index_arrays([tuple(l.weight.shape) for l in lins])[-1][0]          # (2 D, H) positions
    H = made.initial_layer.out_features
    m = made.final_layer.mask.cpu().numpy() != 0
    order = np.argsort(made.initial_layer.degrees.cpu().numpy(), kind="stable")
    hp_train = 256 if H <= 256 else 512
    src = np.zeros((wi.shape[0], hp_train), dtype=np.int32)
    src[:, :H] = np.where(m, wi, 0)[:, order]
    return src


def pack_made_transposed(made, blocks=(1, 2, 3), tri=False):
    """Pack of nf_maf_solve_t: the linear system  v s + J^T g_p(v, g_ld) = g_x  of the implicit backward of the MAF inverse
    (J = dMADE/dx at the solution, g_p the affine transform's parameter cotangent) solved by back-substitution in ONE pass.

    J^T g_p is MADE's input-gradient chain: G4 = Wf^T g_p; G3 = [t1>0] (L11^T G4); G2 = G4 + [h1>0] (L01^T G3); G1 = [t0>0] (L10^T G2);
    G0 = G2 + [h0>0] (L00^T G1); (J^T g_p)_m = (W0^T G0)_m.  Feature m only receives from features > m, a hidden unit of degree d
    only from features >= d and from units of degree >= d: with VIRTUAL feature index f' = D - 1 - m and virtual degree d' = D - d
    this is exactly the structure the incremental inverse walks (a unit of virtual degree d' sees virtual features <= d' - 1, virtual
    feature f' sees units of virtual degree <= f'), for a virtual network whose initial layer is Wf^T (TWO inputs per feature: the
    cotangents of the unconstrained scale and of the shift), whose hidden layers are the transposed hidden layers in reverse order,
    whose final layer is W0^T (ONE row per feature) and whose activations are the ReLU masks of the forward pass.  The tiles are the
    FORWARD plan's tiles in reverse order with the forward (format 0) positions, so the sign bits nf_maf_inverse_h leaves per lane and
    tile are read back as they are.  Record per tile (processing order): A0' [K0'/8][2][32][4] (K0' = 2 x the virtual features
    before the tile, padded to 32) | A1'..A_{NL-1}', AF' [4t'][2][32][4] | W0d' [32][32] (window: 16 steps x 2 inputs) | Wd'[NL-1][32][32]
    | WFd' [32][32] (row j = step j).  No biases.  table: [D, 2D padded to 32, H, Hp, T, 1, NB, 2], per tile
    [dlo', nsteps, K0', rec, mask[16] (virtual step order), forward tile index, regular-8 flag (round 6, `tri` only), 0, 0].
    `tri`: the positions of the FORMAT-1 forward pack (regular tiles / tiles with extras permuted, flows/maf_pack tile_row) -- for the
    masks nf_maf_inverse_h_tri_bits leaves; the solve kernel's sequential part is driven by per-step position masks, so any position
    assignment works."""
    if not supported(made, 2, blocks):
        return None
    base = pack_made(made, blocks=blocks)          # structure checks (degrees, masks) + the forward plan
    if base is None:
        return None
    D = made.initial_layer.in_features
    H = made.initial_layer.out_features
    hid_deg = made.initial_layer.degrees.cpu().numpy()
    lin = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers]
    NB = len(made.blocks)
    NL = 1 + 2 * NB
    order, tiles = plan_tiles(D, hid_deg)
    T = len(tiles)
    Hp = T * TILE
    Dq = (2 * D + 31) // 32 * 32
    # forward (format 0) padded slot of every unit, then the virtual slot: same position, tiles reversed
    fslot = np.zeros(H, dtype=np.int64)
    step_masks = []                                   # per forward tile: position bitmask of every step
    k = 0
    for t, (dlo, ns, steps) in enumerate(tiles):
        b_ = t * TILE
        perm = bool(tri) and (is_regular(steps) or extras_prefix(steps) > 0)
        sm = []
        for g, c in enumerate(steps):
            mk = 0
            for i in range(c):
                pos_ = tile_row(g, i) if perm else b_ - t * TILE
                fslot[order[k]] = t * TILE + pos_
                mk |= 1 << pos_
                b_ += 1
                k += 1
            sm.append(mk)
        step_masks.append(sm)
    vslot = (T - 1 - fslot // TILE) * TILE + fslot % TILE
    mw = lambda l: (l.weight.detach() * l.mask).cpu().numpy().astype(np.float32)
    W0 = mw(lin[0])                                   # (H, D)
    Lh = [mw(l) for l in lin[1:]]                     # L00, L10, L01, L11, ... (H, H): [out, in]
    WF = mw(made.final_layer)                         # (2 D, H), row 2 f + k
    # virtual matrices in padded virtual slot space
    W0v = np.zeros((Hp, Dq), dtype=np.float32)        # [unit][2 f' + k] = WF[2 (D-1-f') + k][unit]
    for fq in range(D):
        for kk in (0, 1):
            W0v[vslot, 2 * fq + kk] = WF[2 * (D - 1 - fq) + kk, :]
    Vh = []
    for kk in range(2 * NB):                          # V_k = (hidden linear 2 NB - 1 - k)^T : [u (receiver = its input)][u' (its output)]
        Lm = Lh[2 * NB - 1 - kk]
        V = np.zeros((Hp, Hp), dtype=np.float32)
        V[np.ix_(vslot, vslot)] = Lm.T
        Vh.append(V)
    WFv = np.zeros((D, Hp), dtype=np.float32)         # [f'][unit] = W0[unit][D-1-f']
    for fq in range(D):
        WFv[fq, vslot] = W0[:, D - 1 - fq]
    chunks = [np.zeros(4, dtype=np.float32)]
    off = 4
    table = np.zeros(TABLE_HDR + TABLE_ENT * T, dtype=np.int32)
    table[0:8] = [D, Dq, H, Hp, T, 1, NB, 2]
    for tq in range(T):
        t = T - 1 - tq
        dlo, ns, steps = tiles[t]
        dloq = D - dlo - ns + 1
        r0, r1 = tq * TILE, (tq + 1) * TILE
        nprev = dloq - 1
        K0 = (2 * nprev + 31) // 32 * 32
        a0 = np.zeros((TILE, K0), dtype=np.float32)
        a0[:, :2 * nprev] = W0v[r0:r1, :2 * nprev]
        rec = [_a_operand(a0)] if K0 else []
        for V in Vh:
            if tq:
                rec.append(_a_operand(V[r0:r1, :r0]))
        fo = np.zeros((TILE, Hp), dtype=np.float32)
        for j in range(ns):
            fo[j] = WFv[dloq + j]
        if tq:
            rec.append(_a_operand(fo[:, :r0]))
        w0d = np.zeros((TILE, 2 * MAX_STEPS), dtype=np.float32)     # window inputs 2 (dlo'-1) ... : 16 steps x 2
        nwin = min(2 * MAX_STEPS, 2 * D - 2 * (dloq - 1))
        w0d[:, :nwin] = W0v[r0:r1, 2 * (dloq - 1):2 * (dloq - 1) + nwin]
        rec.append(w0d.reshape(-1))
        for V in Vh:
            rec.append(np.ascontiguousarray(V[r0:r1, r0:r1]).reshape(-1))
        rec.append(np.ascontiguousarray(fo[:, r0:r1]).reshape(-1))
        rec = np.concatenate(rec)
        e = TABLE_HDR + TABLE_ENT * tq
        table[e + 0], table[e + 1], table[e + 2], table[e + 3] = dloq, ns, K0, off
        # virtual step s' = forward step ns - 1 - s'; positions are the forward ones
        fmask = [np.array([mk], dtype=np.uint64).astype(np.uint32).view(np.int32)[0] for mk in step_masks[t]]
        for sq in range(ns):
            table[e + 4 + sq] = fmask[ns - 1 - sq]
        table[e + 20] = t
        # round 6: REGULAR-8 (format-1 positions, exactly 8 degrees of exactly 4 units): the solve's statically unrolled sequential part
        table[e + 21] = 1 if (tri and ns == FAST_STEPS and all(c == 4 for c in steps)) else 0
        chunks.append(rec)
        off += rec.size
    return np.concatenate(chunks).astype(np.float32), table
