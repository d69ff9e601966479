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
    for b in range(NB):                                      # nets/made.py:196-214
        t = hidden_layer(1 + 2 * b, np.maximum(h, 0.0))
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
    return out[:, :mult * D]


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
