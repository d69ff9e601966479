"""numpy walk-through of MADE's backward pass exactly as csrc/made_bwd.hip performs it, driven by the tables of
flows/made_pack.pack_made_backward: every wave of the input-gradient chain walks its own stream (zero bias group + the transposed
weight's fragments of the item's k-group range), the weight-gradient launch works through its tile list and the reduction scatters
through the row / column maps.  Test infrastructure: validates the packing on CPU against autograd through the dense masked MADE."""
import numpy as np

HDR, ROWS, KG, RING, T = 32, 32, 8, 8, 128


def _rows_from_stream(a, nkg):
    return a.reshape(nkg, 2, ROWS, 4).transpose(2, 0, 1, 3).reshape(ROWS, KG * nkg)


def slot_forward(layers, NB, x, Dp):
    """Pre-activations in slot space from the dense slot-space layers: S[l] (l = 0: h0, 2 b + 1: t_b, 2 b + 2: h_(b+1)), params."""
    B = x.shape[0]
    xin = np.zeros((B, Dp))
    xin[:, :x.shape[1]] = x
    W, _, b = layers[0]
    h = xin @ W.astype(np.float64).T + b
    S = [h]
    for k in range(NB):
        W1, _, b1 = layers[1 + 2 * k]
        W2, _, b2 = layers[2 + 2 * k]
        t = np.maximum(h, 0) @ W1.astype(np.float64).T + b1
        h = h + np.maximum(t, 0) @ W2.astype(np.float64).T + b2
        S += [t, h]
    Wf, _, bf = layers[-1]
    return S, h @ Wf.astype(np.float64).T + bf


def emulate_chain(pack, gp, S):
    """(g_x (B, D), G[l] (B, Hp)) as made_bwd_kernel computes them; S = slot-space pre-activations (their signs = the forward's bits)."""
    blob, table = pack["blob"].astype(np.float64), pack["table"]
    D, Dp, H, Hp, NSB, NB, mult, NC, nfin, total, nitems = [int(v) for v in table[:11]]
    plain = bool(table[13])
    assert blob.size == total and nitems == 2 * NC + 4 * NB + nfin and nfin == (Dp // 32 + 3) // 4
    tab = table[HDR:HDR + 8 * nitems * 4].reshape(8, nitems, 4)
    B = gp.shape[0]
    MD = int(table[12]) if table[12] else mult * D
    gpp = np.zeros((B, NC * Hp))
    gpp[:, :MD] = gp
    pos = [int(table[16 + w]) for w in range(8)]
    start = list(pos)

    def item(w, i, act):
        nkg, rb, kg0 = (int(v) for v in tab[w, i, :3])
        if rb < 0:
            return rb, None
        bias = blob[pos[w]:pos[w] + 1024]
        assert not bias.any()
        pos[w] += 1024
        acc = np.zeros((B, ROWS))
        if nkg:
            assert nkg % 4 == 0 and kg0 % 4 == 0 and KG * (kg0 + nkg) <= act.shape[1]
            W = _rows_from_stream(blob[pos[w]:pos[w] + 256 * nkg], nkg)
            acc = act[:, KG * kg0:KG * (kg0 + nkg)] @ W.T
            pos[w] += 256 * nkg
        return rb, acc

    def layer(i0, act):
        out = np.full((B, Hp), np.nan)
        for w in range(8):
            for s in range(2):
                rb, acc = item(w, i0 + s, act)
                out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        assert not np.isnan(out).any()
        return out

    G = [None] * (2 * NB + 1)
    gh = np.zeros((B, Hp))
    for c in range(NC):
        gh = gh + layer(2 * c, gpp[:, c * Hp:(c + 1) * Hp])
    G[2 * NB] = gh
    if plain:
        assert NB == 1 and (tab[:, 2 * NC:2 * NC + 2, 1] == -1).all()
        gt = gh * (S[1] > 0)
        G[1] = gt
        gh = layer(2 * NC + 2, gt) * (S[0] > 0)
        G[0] = gh
    for k, b in enumerate(range(NB - 1, -1, -1) if not plain else ()):
        i0 = 2 * NC + 4 * k
        gt = layer(i0, gh) * (S[2 * b + 1] > 0)
        G[2 * b + 1] = gt
        gh = gh + layer(i0 + 2, gt) * (S[2 * b] > 0)
        G[2 * b] = gh
    gx = np.full((B, Dp), np.nan)
    seen = {}
    for w in range(8):
        for rd in range(nfin):
            rb, acc = item(w, nitems - nfin + rd, gh)
            if rb >= 0:
                assert rb == (w & 3) + 4 * rd
                if rb in seen:
                    assert np.array_equal(seen[rb], acc)       # the two sample blocks' waves hold the same weights
                seen[rb] = acc
                gx[:, rb * ROWS:(rb + 1) * ROWS] = acc
    assert not np.isnan(gx).any()
    for w in range(8):
        assert np.array_equal(blob[pos[w]:pos[w] + RING * 256], np.resize(blob[start[w]:pos[w]], RING * 256)), w
        assert pos[w] + RING * 256 == (int(table[16 + w + 1]) if w < 7 else total)
    return gx[:, :D], G


def emulate_wgrad(pack, gp, x, G, S):
    """Flat gradient vector as made_wgrad_kernel + made_wgrad_reduce_kernel produce it (one chunk)."""
    wt, sc, mask = pack["wtable"], pack["stable"], pack["mask"]
    ntl, npr = int(wt[0]), int(wt[1])
    assert ntl == pack["ntiles"]
    B = gp.shape[0]
    gpp = np.zeros((B, pack["Mp"]))
    gpp[:, :gp.shape[1]] = gp
    xp = np.zeros((B, pack["Dx"]))
    xp[:, :x.shape[1]] = x
    bases = {0: [gpp], 1: [xp], 2: G, 3: S}
    grads = np.zeros(pack["nflat"])
    written = np.zeros(pack["nflat"], dtype=bool)
    bias_rows = set()
    for t in range(ntl):
        p, m0, n0, wb = (int(v) for v in wt[16 + 8 * npr + 8 * t:16 + 8 * npr + 8 * t + 4])
        dyb, dyl, ldy, xb, xl, ldx, relu = (int(v) for v in wt[16 + 8 * p:16 + 8 * p + 7])
        dY, X = bases[dyb][dyl], bases[xb][xl]
        assert dY.shape[1] == ldy and X.shape[1] == ldx and m0 + T <= ldy and n0 + T <= ldx
        Xn = np.maximum(X, 0) if relu else X
        tile = dY[:, m0:m0 + T].T @ Xn[:, n0:n0 + T]
        woff, ldw, boff, ro, co, bo = (int(v) for v in sc[8 * p:8 * p + 6])
        rmap, cmap = sc[ro + m0:ro + m0 + T], sc[co + n0:co + n0 + T]
        bmap = sc[bo + m0:bo + m0 + T] if bo else rmap
        for mm in range(T):
            if rmap[mm] < 0:
                continue
            for nn in range(T):
                if cmap[nn] >= 0:
                    dst = woff + int(rmap[mm]) * ldw + int(cmap[nn])
                    if mask[dst]:
                        assert not written[dst]
                        grads[dst] = tile[mm, nn]
                        written[dst] = True
            if wb:
                assert (p, m0 + mm) not in bias_rows
                bias_rows.add((p, m0 + mm))
                grads[boff + int(bmap[mm])] = dY[:, m0 + mm].sum()
                written[boff + int(bmap[mm])] = True
    assert np.array_equal(written, mask != 0)         # every unmasked entry and every bias exactly once, nothing else
    return grads
