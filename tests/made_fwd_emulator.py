"""numpy walk-through of the one-launch MADE forward exactly as csrc/made_fwd.hip performs it, driven by the packed blob / table
of flows/made_pack.py.  Test infrastructure: validates the packing (slot order, A-operand streams, prefix k-group counts, biases)
on CPU against the dense masked MADE."""
import numpy as np

HDR, ROWS, KG = 32, 32, 8


def _rows_from_stream(a, nkg):
    return a.reshape(nkg, 2, ROWS, 4).transpose(2, 0, 1, 3).reshape(ROWS, KG * nkg)


def emulate_forward(blob, table, x):
    """MADE output (B, mult D) in the reference's row order, float64 arithmetic on the packed float32 weights."""
    blob = blob.astype(np.float64)
    D, Dp, H, Hp, NSB, NB, mult, NFB, nlayers, total = [int(v) for v in table[:10]]
    assert blob.size == total and nlayers == 2 * NB + 2 and Hp == 256 * NSB
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    xin = np.zeros((B, Dp))
    xin[:, :D] = x

    def layer(l, act, nrb):
        d = table[int(table[16 + l]):int(table[16 + l]) + 4 * nrb].reshape(nrb, 4)
        out = np.zeros((B, nrb * ROWS))
        for rb in range(nrb):
            a_off, nkg, b_off = int(d[rb, 0]), int(d[rb, 1]), int(d[rb, 2])
            acc = np.tile(blob[b_off:b_off + ROWS], (B, 1))
            if nkg:
                W = _rows_from_stream(blob[a_off:a_off + ROWS * KG * nkg], nkg)
                acc = acc + act[:, :KG * nkg] @ W.T
            out[:, rb * ROWS:(rb + 1) * ROWS] = acc
        return out

    nrb = Hp // ROWS
    h = layer(0, xin, nrb)                                   # initial layer: raw h0
    for b in range(NB):                                      # nets/made.py:196-214
        t = layer(1 + 2 * b, np.maximum(h, 0.0), nrb)
        h = h + layer(2 + 2 * b, np.maximum(t, 0.0), nrb)
    out = layer(2 * NB + 1, h, NFB)                          # the final layer sees the raw block output (:303-304)
    return out[:, :mult * D]


def work_fraction(table):
    """MFMA k-groups the schedule executes / the dense count (padded sizes)."""
    D, Dp, H, Hp, NSB, NB, mult, NFB, nlayers, total = [int(v) for v in table[:10]]
    done = dense = 0
    for l in range(nlayers):
        nrb = NFB if l == nlayers - 1 else Hp // ROWS
        K = Dp if l == 0 else Hp
        d = table[int(table[16 + l]):int(table[16 + l]) + 4 * nrb].reshape(nrb, 4)
        done += int(d[:, 1].sum())
        dense += nrb * (K // KG)
    return done / dense
