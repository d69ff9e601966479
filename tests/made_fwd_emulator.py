"""numpy walk-through of the one-launch MADE forward exactly as csrc/made_fwd.hip performs it, driven by the packed blob / table
of flows/made_pack.py: every wave walks its own stream (bias group, then the A fragments of the item) in the kernel's item order.
Test infrastructure: validates the packing (slot order, per-wave streams, prefix k-group counts, biases, the wrap-around copy) on
CPU against the dense masked MADE."""
import numpy as np

HDR, ROWS, KG, RING = 32, 32, 8, 8


def _rows_from_stream(a, nkg):
    return a.reshape(nkg, 2, ROWS, 4).transpose(2, 0, 1, 3).reshape(ROWS, KG * nkg)


def _bias_from_group(g):
    g = g.reshape(4, 2, ROWS, 4)
    assert np.array_equal(g, np.broadcast_to(g[:, :, :1, :], g.shape))      # the same for every lane of a half
    return g[:, :, 0, :].reshape(32)                                          # index 8 q + 4 hh + i


def emulate_forward(blob, table, x):
    """MADE output (B, mult D) in the reference's row order, float64 arithmetic on the packed float32 weights."""
    blob = blob.astype(np.float64)
    D, Dp, H, Hp, NSB, NB, mult, NFB, nrounds, total, nitems = [int(v) for v in table[:11]]
    assert blob.size == total and nitems == 2 * (1 + 2 * NB) + 2 * nrounds and Hp == 256 * NSB and Dp % 32 == 0
    tab = table[HDR:HDR + 8 * nitems * 2].reshape(8, nitems, 2)
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    xin = np.zeros((B, Dp))
    xin[:, :D] = x
    pos = [int(table[16 + w]) for w in range(8)]
    start = list(pos)

    def item(w, i, act):
        nkg, rb = int(tab[w, i, 0]), int(tab[w, i, 1])
        if rb < 0:
            return rb, None
        acc = np.tile(_bias_from_group(blob[pos[w]:pos[w] + 4 * 256]), (B, 1))
        pos[w] += 4 * 256
        if nkg:
            assert nkg % 4 == 0 and KG * nkg <= act.shape[1]
            W = _rows_from_stream(blob[pos[w]:pos[w] + 256 * nkg], nkg)
            acc = acc + act[:, :KG * nkg] @ W.T
            pos[w] += 256 * nkg
        return rb, acc

    def hidden_layer(l, act):
        out = np.full((B, Hp), np.nan)
        for w in range(8):
            for s in range(2):
                rb, acc = item(w, 2 * l + s, act)
                out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        assert not np.isnan(out).any()                       # every row-block has exactly one owner
        return out

    h = hidden_layer(0, xin)                                 # initial layer: raw h0
    plain = bool(table[13])
    for b in range(NB):                                      # nets/made.py:196-214
        t = hidden_layer(1 + 2 * b, np.maximum(h, 0.0))
        if plain:                                            # x -> W0 -> relu -> W1 -> relu -> Wf: the final layer sees relu(t)
            assert NB == 1 and (tab[:, 4:6, 1] == -1).all()
            h = np.maximum(t, 0.0)
            break
        h = h + hidden_layer(2 + 2 * b, np.maximum(t, 0.0))
    out = np.full((B, NFB * ROWS), np.nan)
    for r in range(nrounds):                                 # the final layer sees the raw block output (:303-304)
        for w in range(8):
            for s in range(2):
                rb, acc = item(w, 2 * (1 + 2 * NB) + 2 * r + s, h)
                if rb >= 0:
                    out[:, rb * ROWS:(rb + 1) * ROWS] = acc
    assert not np.isnan(out).any()
    for w in range(8):                                       # the wrap-around copy behind every stream
        n = pos[w] - start[w]
        assert np.array_equal(blob[pos[w]:pos[w] + RING * 256], np.resize(blob[start[w]:pos[w]], RING * 256)), w
        assert pos[w] + RING * 256 == (int(table[16 + w + 1]) if w < 7 else total) and n > 0
    return out[:, :(int(table[12]) if table[12] else mult * D)]


def work_per_wave(table):
    """MFMAs per wave and 64-row tile (a k-group of a row-block = 4 MFMAs per 32-sample block)."""
    NSB, NB, nitems = int(table[4]), int(table[5]), int(table[10])
    tab = table[HDR:HDR + 8 * nitems * 2].reshape(8, nitems, 2)
    nh = 2 * (1 + 2 * NB)                                   # hidden items: NSB sample blocks each; final items: one
    return 4 * (NSB * tab[:, :nh, 0].sum(axis=1) + tab[:, nh:, 0].sum(axis=1))


def work_fraction(table):
    """MFMAs the schedule executes / the dense count (padded sizes)."""
    D, Dp, H, Hp, NSB, NB, mult, NFB, nrounds, total, nitems = [int(v) for v in table[:11]]
    dense = (Hp // ROWS) * (Dp // KG) + 2 * NB * (Hp // ROWS) * (Hp // KG) + NFB * (Hp // KG)
    return int(work_per_wave(table).sum()) / (8 * dense)


def emulate_forward_spline(blob, table, x):
    """The spline variant (flows/made_pack.py spline=True): (B, D, 24) parameter lists as the kernel's lanes hold them (widths /
    heights carry log2(e)), float64 arithmetic on the packed float32 weights."""
    blob = blob.astype(np.float64)
    D, Dp, H, Hp, NSB, NB, mult, G, nfi, total, nitems, spl = [int(v) for v in table[:12]]
    assert spl == 1 and mult == 23 and blob.size == total and nitems == 2 * (1 + 2 * NB) + nfi
    tab = table[HDR:HDR + 8 * nitems * 2].reshape(8, nitems, 2)
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    xin = np.zeros((B, Dp))
    xin[:, :D] = x
    pos = [int(table[16 + w]) for w in range(8)]
    start = list(pos)

    def hidden_layer(l, act):
        out = np.full((B, Hp), np.nan)
        for w in range(8):
            for s in range(2):
                nkg, rb = int(tab[w, 2 * l + s, 0]), int(tab[w, 2 * l + s, 1])
                acc = np.tile(_bias_from_group(blob[pos[w]:pos[w] + 1024]), (B, 1))
                pos[w] += 1024
                if nkg:
                    acc = acc + act[:, :KG * nkg] @ _rows_from_stream(blob[pos[w]:pos[w] + 256 * nkg], nkg).T
                    pos[w] += 256 * nkg
                out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        assert not np.isnan(out).any()
        return out

    h = hidden_layer(0, xin)
    for b in range(NB):
        t = hidden_layer(1 + 2 * b, np.maximum(h, 0.0))
        h = h + hidden_layer(2 + 2 * b, np.maximum(t, 0.0))
    prm = np.full((B, 4 * G, 24), np.nan)
    nh = 2 * (1 + 2 * NB)
    seen = set()
    for w in range(8):
        for j in range(nfi):
            nkg, g = int(tab[w, nh + j, 0]), int(tab[w, nh + j, 1])
            if g < 0:
                continue
            assert g not in seen and nkg % 4 == 0
            seen.add(g)
            acc = np.zeros((3, B, ROWS))
            for r3 in range(3):
                acc[r3] = np.tile(_bias_from_group(blob[pos[w]:pos[w] + 1024]), (B, 1))
                pos[w] += 1024
            if nkg:
                frag = blob[pos[w]:pos[w] + 3 * 256 * nkg].reshape(nkg, 3, 256)
                pos[w] += 3 * 256 * nkg
                for r3 in range(3):
                    acc[r3] += h[:, :KG * nkg] @ _rows_from_stream(np.ascontiguousarray(frag[:, r3]).reshape(-1), nkg).T
            for r3 in range(3):
                for rho in range(ROWS):
                    q, hh, i = rho >> 3, (rho >> 2) & 1, rho & 3
                    v = 16 * r3 + 4 * q + i
                    prm[:, 4 * g + 2 * hh + v // 24, v % 24] = acc[r3][:, rho]
    assert seen == set(range(G))
    for w in range(8):
        assert np.array_equal(blob[pos[w]:pos[w] + RING * 256], np.resize(blob[start[w]:pos[w]], RING * 256)), w
        assert pos[w] + RING * 256 == (int(table[16 + w + 1]) if w < 7 else total)
    return prm[:, :D]
