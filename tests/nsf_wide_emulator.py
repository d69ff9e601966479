"""numpy walk-through of the NSF coupling layer's conditioner exactly as csrc/nsf_wide.hip walks the packed streams of
flows/nsf_wide_pack.py (hidden items per wave, final layer in groups of 8 / 4 / 2 transform features (4 / 8 / 16 bins) = 3 row-blocks
whose accumulator registers are the lane's parameter lists).  Test infrastructure: pins the packing on CPU against the dense ResidualNet."""
import numpy as np

HDR, ROWS, KG, RING, M, MP = 32, 32, 8, 8, 23, 24


def _rows(a, nkg):
    return a.reshape(nkg, 2, ROWS, 4).transpose(2, 0, 1, 3).reshape(ROWS, KG * nkg)


def _bias(g):
    g = g.reshape(4, 2, ROWS, 4)
    assert np.array_equal(g, np.broadcast_to(g[:, :, :1, :], g.shape))
    return g[:, :, 0, :].reshape(32)


def emulate_conditioner(blob, table, x, direction=0):
    """((B, nT, 3 K) parameter lists as the kernel's lanes hold them (widths / heights still carry log2(e) / sqrt(hidden)), LU output
    (B, D) or None), from full rows x (B, D).  With a fused LU: density = LU(x) first and the conditioner sees ITS output; sampling =
    the LU item comes last in the streams and is applied to x here only to check its packing."""
    blob = blob.astype(np.float64)
    D, Dp, H, Hp, NB, nI, nT, par_i, par_t, G, nfi, total, nhi, has_lu, TR, PI = [int(v) for v in table[:16]]
    assert TR == (128 if Hp == 128 else 64) and PI % 32 == 0 and (Dp - PI) % 32 == 0 and PI >= nI and Dp - PI >= nT
    K = int(table[24]) or 8
    MP = 3 * K                       # slots per feature; a lane-half holds FPL = 48 / MP features per group
    FPL = 48 // MP
    FPG = 2 * FPL
    assert G == (nT + FPG - 1) // FPG
    nhl = 1 + 2 * NB
    nitems = nhl * nhi + nfi + has_lu
    base = 1 if (has_lu and direction == 0) else 0
    tab = table[HDR:HDR + 8 * nitems * 3].reshape(8, nitems, 3)
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    xin = np.zeros((B, Dp))                                   # the tile in position order: identity features, then transform features
    xin[:, :nI] = x[:, par_i::2]
    xin[:, PI:PI + nT] = x[:, par_t::2]
    pos = [int(table[16 + w]) for w in range(8)]
    start = list(pos)

    def lu_stage(idx, act):
        out = np.zeros((B, Dp))
        for w in range(8):
            nkg, rb, sb0 = [int(v) for v in tab[w, idx]]
            if rb < 0:
                continue
            assert rb == (w & 3) and sb0 == (w >> 2) * (TR // 64) and KG * nkg == Dp
            acc = np.tile(_bias(blob[pos[w]:pos[w] + 1024]), (B, 1)) + act @ _rows(blob[pos[w] + 1024:pos[w] + 1024 + 256 * nkg], nkg).T
            pos[w] += 1024 + 256 * nkg
            out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        return out

    def to_columns(t):
        out = np.zeros((B, D))
        out[:, par_i::2] = t[:, :nI]
        out[:, par_t::2] = t[:, PI:PI + nT]
        return out

    lu_out = None
    if has_lu and direction == 0:
        xin = lu_stage(0, xin)
        lu_out = to_columns(xin)

    def hidden_layer(l, act):
        out = np.full((B, Hp), np.nan)
        for w in range(8):
            for i in range(nhi):
                nkg, rb, sb0 = [int(v) for v in tab[w, base + l * nhi + i]]
                acc = np.tile(_bias(blob[pos[w]:pos[w] + 1024]), (B, 1))
                pos[w] += 1024
                # the initial layer contracts over the identity positions; the others over the hidden width rounded up to 32 (round 5:
                # not over Hp -- a 192-wide network does not pay for its padding to 256 in K)
                assert KG * nkg == (PI if l == 0 else (H + 31) // 32 * 32) and KG * nkg <= act.shape[1]
                assert l == 0 or not act[:, KG * nkg:].any()                # what the shorter k-loop skips is padding: exact zeros
                acc = acc + act[:, :KG * nkg] @ _rows(blob[pos[w]:pos[w] + 256 * nkg], nkg).T
                pos[w] += 256 * nkg
                # (a wave computes only its sample blocks; the emulation ignores the split: every owner must agree)
                prev = out[:, rb * ROWS:(rb + 1) * ROWS]
                assert np.isnan(prev).all() or np.array_equal(prev, acc)
                out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        assert not np.isnan(out).any()
        return out

    h = hidden_layer(0, xin)
    for b in range(NB):
        t = hidden_layer(1 + 2 * b, np.maximum(h, 0.0))
        h = h + hidden_layer(2 + 2 * b, np.maximum(t, 0.0))
    prm = np.zeros((B, FPG * G, MP))
    seen = set()
    for w in range(8):
        for j in range(nfi):
            nkg, g, sb0 = [int(v) for v in tab[w, base + nhl * nhi + j]]
            if g < 0:
                continue
            assert (g, sb0) not in seen and sb0 in range(0, TR // 32, 2)
            seen.add((g, sb0))
            acc = np.zeros((3, B, ROWS))
            for r3 in range(3):
                acc[r3] = np.tile(_bias(blob[pos[w]:pos[w] + 1024]), (B, 1))
                pos[w] += 1024
            frag = blob[pos[w]:pos[w] + 3 * 256 * nkg].reshape(nkg, 3, 256)
            pos[w] += 3 * 256 * nkg
            assert KG * nkg == (H + 31) // 32 * 32 and not h[:, KG * nkg:].any()
            for r3 in range(3):
                acc[r3] += h[:, :KG * nkg] @ _rows(np.ascontiguousarray(frag[:, r3]).reshape(-1), nkg).T
            for r3 in range(3):
                for rho in range(ROWS):
                    q, hh, i = rho >> 3, (rho >> 2) & 1, rho & 3
                    v = 16 * r3 + 4 * q + i
                    prm[:, FPG * g + FPL * hh + v // MP, v % MP] = acc[r3][:, rho]
    assert seen == {(g, sb0) for g in range(G) for sb0 in range(0, TR // 32, 2)}      # every (group, sample-block pair) has one owner
    if has_lu and direction == 1:
        lu_out = to_columns(lu_stage(nitems - 1, xin))
    for w in range(8):
        n = pos[w] - start[w]
        assert np.array_equal(blob[pos[w]:pos[w] + RING * 256], np.resize(blob[start[w]:pos[w]], RING * 256)), w
        assert pos[w] + RING * 256 == (int(table[16 + w + 1]) if w < 7 else total) and n > 0
    return prm[:, :nT], lu_out
