"""numpy walk-through of the incremental MAF inverse exactly as csrc/maf_inverse.hip performs it, driven by the packed
blob/table of flows/maf_pack.py.  Test infrastructure: validates the packing and the schedule on CPU."""
import numpy as np

TILE, MAX_STEPS, HDR, ENT = 32, 16, 8, 24


def _from_a_operand(a, K):
    return a.reshape(K // 8, 2, TILE, 4).transpose(2, 0, 1, 3).reshape(TILE, K)


def emulate_inverse(blob, table, z, element=None, return_scratch=False):
    """`element(params (B, mult), z_f (B,)) -> (x_f, logabsdet_f)`: the element-wise inverse of the rows layout (one
    final-layer block per feature, nf_arnsf_inverse); None = the affine layout of nf_maf_inverse."""
    z = np.asarray(z, dtype=np.float64)
    B = z.shape[0]
    D, Dp, H, Hp, T = [int(v) for v in table[:5]]
    if element is not None:
        return _emulate_rows(blob.astype(np.float64), table, z, element)
    blob = blob.astype(np.float64)
    NB = int(table[6]) if int(table[6]) > 0 else 2      # residual blocks; NL = 1 + 2 NB hidden layers
    NL = 1 + 2 * NB
    x = np.zeros((B, Dp))
    S = np.zeros((NL, B, Hp))
    ld = np.zeros(B)

    def finish(us, sh, zf):
        scale = 1.0 / (1.0 + np.exp(-(us + 2.0))) + 1e-3
        return (zf - sh) / scale, -np.log(scale)

    x[:, 0], d = finish(blob[0], blob[1], z[:, 0])
    ld += d
    for t in range(T):
        e = HDR + ENT * t
        dlo, ns, K0, off = [int(v) for v in table[e:e + 4]]
        masks = table[e + 4:e + 4 + MAX_STEPS].view(np.uint32)
        Kh = TILE * t
        A0 = _from_a_operand(blob[off:off + K0 * TILE], K0) if K0 else np.zeros((TILE, 0)); off += K0 * TILE
        Ah = []
        for _ in range(NL):  # A1..A_{NL-1}, AF
            Ah.append(_from_a_operand(blob[off:off + Kh * TILE], Kh) if Kh else np.zeros((TILE, 0))); off += Kh * TILE
        bias = blob[off:off + NL * TILE].reshape(NL, TILE); off += NL * TILE
        biasF = blob[off:off + TILE]; off += TILE
        if int(table[7]) == 1 and int(table[e + 20]) in (1, 2):
            # format 1, REGULAR tile: the statically unrolled triangular steps of maf_inverse_h.hip (fast path), lane-half by
            # lane-half and register by register, reading the record exactly where the kernel reads it
            pre = np.zeros((NL, B, TILE))
            pre[0] = x[:, :K0] @ A0.T + bias[0]
            for l in range(1, NL):
                pre[l] = S[l - 1][:, :Kh] @ Ah[l - 1].T + bias[l]
            preF = S[NL - 1][:, :Kh] @ Ah[NL - 1].T + biasF
            row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
            HALF = 4 * (20 + (NL - 1) * 72 + 36)
            P = np.zeros((NL, 2, 16, B))                 # P[l][half][reg] = accumulator register `reg` of lane-half `half`
            for l in range(NL):
                for h in (0, 1):
                    for r in range(16):
                        P[l, h, r] = pre[l][:, row(r, h)]
            PF = np.stack([np.stack([preF[:, row(r, h)] for r in range(16)]) for h in (0, 1)])
            half = [blob[off + h * HALF:off + (h + 1) * HALF].reshape(-1, 4) for h in (0, 1)]
            # kind 2 (regular with extras): the fifth unit of step g < m lives in lane-half g & 1, register 14 + (g >> 1); per half an
            # extra region X1 [NL-1][7][2] | X2 [NL-1][4][3] | XW [4] | XF [7] (float4 units) behind the two regular halves
            kind2 = int(table[e + 20]) == 2
            m_x = int(table[e + 21])
            XH = 4 * ((NL - 1) * (7 * 2 + 4 * 3) + 4 + 7)
            xh = [blob[off + 2 * HALF + h * XH:off + 2 * HALF + (h + 1) * XH].reshape(-1, 4) for h in (0, 1)] if kind2 else None
            oX1, oX2, oXW, oXF = 0, (NL - 1) * 14, (NL - 1) * 14 + (NL - 1) * 12, (NL - 1) * 14 + (NL - 1) * 12 + 4
            xg = np.zeros((18, B))
            xg[0] = x[:, dlo - 1]
            o0, od, of = 0, 20, 20 + (NL - 1) * 72
            for g in range(ns):
                b0 = o0 + sum(gg // 2 + 1 for gg in range(g))
                for h in (0, 1):                          # initial layer: each half finishes its own two targets
                    a = [P[0, h, 2 * g].copy(), P[0, h, 2 * g + 1].copy()]
                    for q in range(g // 2 + 1):
                        w = half[h][b0 + q]
                        a[0] += w[0] * xg[2 * q] + w[1] * xg[2 * q + 1]
                        a[1] += w[2] * xg[2 * q] + w[3] * xg[2 * q + 1]
                    for i in (0, 1):
                        P[2, h, 2 * g + i] += a[i]
                        P[0, h, 2 * g + i] = np.maximum(a[i], 0)
                if kind2 and g < m_x:                     # the extra target's initial layer: its own half only
                    ho, rx = g & 1, 14 + (g >> 1)
                    w = xh[ho][oXW + g]
                    ax = P[0, ho, rx] + w[0] * xg[0] + w[1] * xg[1] + w[2] * xg[2] + w[3] * xg[3]
                    P[2, ho, rx] += ax
                    P[0, ho, rx] = np.maximum(ax, 0)

                def product(l):
                    """partials of the four targets over each half's registers 0..2g+1 of layer l, then the two exchanges:
                    returns tot[half][i] for the half's own targets (registers 2g, 2g+1)"""
                    bd = od + l * 72 + g * (g + 1)
                    part = np.zeros((2, 4, B))
                    for h in (0, 1):
                        for q in range(g + 1):
                            for j in (0, 1):
                                w = half[h][bd + 2 * q + j]
                                part[h, 2 * j] += w[0] * P[l, h, 2 * q] + w[1] * P[l, h, 2 * q + 1]
                                part[h, 2 * j + 1] += w[2] * P[l, h, 2 * q] + w[3] * P[l, h, 2 * q + 1]
                    totx = None
                    if kind2:
                        for h in (0, 1):                  # pair 7 (the extras' registers 14, 15) -> the four regular targets
                            for j in (0, 1):
                                w = xh[h][oX1 + (l * 7 + g) * 2 + j]
                                part[h, 2 * j] += w[0] * P[l, h, 14] + w[1] * P[l, h, 15]
                                part[h, 2 * j + 1] += w[2] * P[l, h, 14] + w[3] * P[l, h, 15]
                        if g < m_x:                       # the extra target over pairs 0..g and 7 of both halves
                            totx = np.zeros(B)
                            for h in (0, 1):
                                w = xh[h][oX2 + (l * 4 + g) * 3:oX2 + (l * 4 + g) * 3 + 3].reshape(-1)
                                srcs = [P[l, h, r] for q in range(g + 1) for r in (2 * q, 2 * q + 1)] + [P[l, h, 14], P[l, h, 15]]
                                for k, sv in enumerate(srcs):
                                    totx = totx + w[k] * sv
                    # v_permlane32_swap(q0, q2) + add: half 0 <- target 0, half 1 <- target 2; (q1, q3): targets 1 / 3
                    return [[part[0, 0] + part[1, 0], part[0, 1] + part[1, 1]], [part[0, 2] + part[1, 2], part[0, 3] + part[1, 3]]], totx
                hx, rxx = g & 1, 14 + (g >> 1)
                for b in range(NB):
                    tot, totx = product(2 * b)
                    for h in (0, 1):
                        for i in (0, 1):
                            P[2 * b + 1, h, 2 * g + i] = np.maximum(P[2 * b + 1, h, 2 * g + i] + tot[h][i], 0)
                    if totx is not None:
                        P[2 * b + 1, hx, rxx] = np.maximum(P[2 * b + 1, hx, rxx] + totx, 0)
                    tot, totx = product(2 * b + 1)
                    for h in (0, 1):
                        for i in (0, 1):
                            hn = P[2 * b + 2, h, 2 * g + i] + tot[h][i]
                            if b + 1 < NB:
                                P[2 * b + 4, h, 2 * g + i] += hn
                                P[2 * b + 2, h, 2 * g + i] = np.maximum(hn, 0)
                            else:
                                P[2 * b + 2, h, 2 * g + i] = hn
                    if totx is not None:
                        hn = P[2 * b + 2, hx, rxx] + totx
                        if b + 1 < NB:
                            P[2 * b + 4, hx, rxx] += hn
                            P[2 * b + 2, hx, rxx] = np.maximum(hn, 0)
                        else:
                            P[2 * b + 2, hx, rxx] = hn
                bf_ = of + g * (g + 1) // 2
                pu, ps = np.zeros((2, B)), np.zeros((2, B))
                for h in (0, 1):
                    for q in range(g + 1):
                        w = half[h][bf_ + q]
                        pu[h] += w[0] * P[NL - 1, h, 2 * q] + w[1] * P[NL - 1, h, 2 * q + 1]
                        ps[h] += w[2] * P[NL - 1, h, 2 * q] + w[3] * P[NL - 1, h, 2 * q + 1]
                if kind2:
                    for h in (0, 1):
                        w = xh[h][oXF + g]
                        pu[h] += w[0] * P[NL - 1, h, 14] + w[1] * P[NL - 1, h, 15]
                        ps[h] += w[2] * P[NL - 1, h, 14] + w[3] * P[NL - 1, h, 15]
                us = pu[0] + pu[1] + PF[0, g]             # half 0 holds the scale row in register g, half 1 the shift row
                sh = ps[0] + ps[1] + PF[1, g]
                xn, d = finish(us, sh, z[:, dlo + g])
                ld += d
                x[:, dlo + g] = xn
                xg[g + 1] = xn
            for l in range(NL):
                for h in (0, 1):
                    for r in range(16):
                        S[l][:, TILE * t + row(r, h)] = P[l, h, r]
            continue
        W0d = blob[off:off + TILE * MAX_STEPS].reshape(TILE, MAX_STEPS); off += TILE * MAX_STEPS
        Wd = blob[off:off + (NL - 1) * TILE * TILE].reshape(NL - 1, TILE, TILE); off += (NL - 1) * TILE * TILE
        WFd = blob[off:off + TILE * TILE].reshape(TILE, TILE); off += TILE * TILE
        pre = np.zeros((NL, B, TILE))
        pre[0] = x[:, :K0] @ A0.T + bias[0]
        for l in range(1, NL):
            pre[l] = S[l - 1][:, :Kh] @ Ah[l - 1].T + bias[l]
        preF = S[NL - 1][:, :Kh] @ Ah[NL - 1].T + biasF
        xg = np.zeros((B, MAX_STEPS + 1))
        xg[:, 0] = x[:, dlo - 1]
        for s in range(ns):
            units = [u for u in range(TILE) if (int(masks[s]) >> u) & 1]
            # initial layer; its output h0 is folded into the pre-activation of the first block's second linear
            for u in units:
                h = pre[0][:, u] + xg[:, :MAX_STEPS] @ W0d[u]
                pre[2][:, u] += h
                pre[0][:, u] = np.maximum(h, 0)
            for b in range(NB):
                for u in units:          # t_b = L0_b(relu(h_b))
                    pre[2 * b + 1][:, u] = np.maximum(pre[2 * b + 1][:, u] + pre[2 * b] @ Wd[2 * b][u], 0)
                for u in units:          # h_{b+1} = h_b + L1_b(relu(t_b))
                    hn = pre[2 * b + 2][:, u] + pre[2 * b + 1] @ Wd[2 * b + 1][u]
                    if b + 1 < NB:
                        pre[2 * b + 4][:, u] += hn
                        pre[2 * b + 2][:, u] = np.maximum(hn, 0)
                    else:
                        pre[2 * b + 2][:, u] = hn          # the final layer's input (made.py:304: no activation before it)
            us = preF[:, 2 * s] + pre[NL - 1] @ WFd[2 * s]
            sh = preF[:, 2 * s + 1] + pre[NL - 1] @ WFd[2 * s + 1]
            xn, d = finish(us, sh, z[:, dlo + s])
            ld += d
            x[:, dlo + s] = xn
            xg[:, s + 1] = xn
        for l in range(NL):
            S[l][:, TILE * t:TILE * (t + 1)] = pre[l]
    if return_scratch:             # S (NL, B, Hp): the published activations (the inputs of MADE's linears) in position order
        return x[:, :D], ld, S
    return x[:, :D], ld


def _emulate_rows(blob, table, z, element):
    B = z.shape[0]
    D, Dp, H, Hp, T, mult = [int(v) for v in table[:6]]
    x = np.zeros((B, Dp))
    S = np.zeros((5, B, Hp))
    x[:, 0], ld = element(np.broadcast_to(blob[:mult], (B, mult)), z[:, 0])
    ld = np.array(ld, dtype=np.float64)
    for t in range(T):
        e = HDR + ENT * t
        dlo, ns, K0, off = [int(v) for v in table[e:e + 4]]
        masks = table[e + 4:e + 4 + MAX_STEPS].view(np.uint32)
        Kh = TILE * t

        def a_block(K):
            nonlocal off
            a = _from_a_operand(blob[off:off + K * TILE], K) if K else np.zeros((TILE, 0))
            off += K * TILE
            return a
        A0 = a_block(K0)
        Ah = [a_block(Kh) for _ in range(4)]
        AF = [a_block(Kh) for _ in range(ns)]
        bias = blob[off:off + 5 * TILE].reshape(5, TILE); off += 5 * TILE
        W0d = blob[off:off + TILE * MAX_STEPS].reshape(TILE, MAX_STEPS); off += TILE * MAX_STEPS
        Wd = blob[off:off + 4 * TILE * TILE].reshape(4, TILE, TILE); off += 4 * TILE * TILE
        biasF = blob[off:off + ns * TILE].reshape(ns, TILE); off += ns * TILE
        WFd = blob[off:off + ns * mult * TILE].reshape(ns, mult, TILE); off += ns * mult * TILE
        pre = np.zeros((5, B, TILE))
        pre[0] = x[:, :K0] @ A0.T + bias[0]
        for l in range(1, 5):
            pre[l] = S[l - 1][:, :Kh] @ Ah[l - 1].T + bias[l]
        xg = np.zeros((B, MAX_STEPS + 1))
        xg[:, 0] = x[:, dlo - 1]
        for s in range(ns):
            units = [u for u in range(TILE) if (int(masks[s]) >> u) & 1]
            for u in units:
                h = pre[0][:, u] + xg[:, :MAX_STEPS] @ W0d[u]
                pre[2][:, u] += h
                pre[0][:, u] = np.maximum(h, 0)
            for u in units:
                pre[1][:, u] = np.maximum(pre[1][:, u] + pre[0] @ Wd[0][u], 0)
            for u in units:
                h1 = pre[2][:, u] + pre[1] @ Wd[1][u]
                pre[4][:, u] += h1
                pre[2][:, u] = np.maximum(h1, 0)
            for u in units:
                pre[3][:, u] = np.maximum(pre[3][:, u] + pre[2] @ Wd[2][u], 0)
            for u in units:
                pre[4][:, u] = pre[4][:, u] + pre[3] @ Wd[3][u]
            prm = (S[4][:, :Kh] @ AF[s].T + biasF[s])[:, :mult] + pre[4] @ WFd[s].T
            xn, d = element(prm, z[:, dlo + s])
            ld = ld + d
            x[:, dlo + s] = xn
            xg[:, s + 1] = xn
        for l in range(5):
            S[l][:, TILE * t:TILE * (t + 1)] = pre[l]
    return x[:, :D], ld


def emulate_solve_t(blob, table, x, params, gx, gld, masks, return_scratch=False):
    """numpy walk-through of csrc/maf_solve_t.hip (format 2 of flows/maf_pack.pack_made_transposed): the one-pass back-substitution of
    v s + J^T g_p(v, g_ld) = g_x.  masks[k - 1] (B, Hp) = the ReLU mask of VIRTUAL layer k = 1 .. 2 NB in virtual slot order (the sign of
    forward layer 2 NB - k's pre-activation of the unit in that slot)."""
    blob = blob.astype(np.float64)
    B = x.shape[0]
    D, Dq, H, Hp, T = [int(v) for v in table[:5]]
    assert int(table[7]) == 2 and int(table[5]) == 1
    NB = int(table[6])
    NL = 1 + 2 * NB
    X = np.zeros((B, Dq))                 # virtual inputs: (g_us, g_sh) of virtual feature f' at 2 f', 2 f' + 1
    S = np.zeros((NL, B, Hp))
    v = np.zeros((B, D))

    def finish(gxm, f):                   # real feature f
        sg = 1.0 / (1.0 + np.exp(-(params[:, 2 * f] + 2.0)))
        scale = sg + 1e-3
        vf = (gx[:, f] - gxm) / scale
        return vf, (vf * x[:, f] + gld / scale) * sg * (1.0 - sg), vf

    v[:, D - 1], X[:, 0], X[:, 1] = finish(0.0, D - 1)
    for t in range(T):
        e = HDR + ENT * t
        dlo, ns, K0, off = [int(q) for q in table[e:e + 4]]
        msk = table[e + 4:e + 4 + MAX_STEPS].view(np.uint32)
        Kh = TILE * t
        A0 = _from_a_operand(blob[off:off + K0 * TILE], K0) if K0 else np.zeros((TILE, 0)); off += K0 * TILE
        Ah = []
        for _ in range(NL):
            Ah.append(_from_a_operand(blob[off:off + Kh * TILE], Kh) if Kh else np.zeros((TILE, 0))); off += Kh * TILE
        W0d = blob[off:off + TILE * 2 * MAX_STEPS].reshape(TILE, 2 * MAX_STEPS); off += TILE * 2 * MAX_STEPS
        Wd = blob[off:off + (NL - 1) * TILE * TILE].reshape(NL - 1, TILE, TILE); off += (NL - 1) * TILE * TILE
        WFd = blob[off:off + TILE * TILE].reshape(TILE, TILE); off += TILE * TILE
        pre = np.zeros((NL, B, TILE))
        pre[0] = X[:, :K0] @ A0.T
        for l in range(1, NL):
            pre[l] = S[l - 1][:, :Kh] @ Ah[l - 1].T
        preF = S[NL - 1][:, :Kh] @ Ah[NL - 1].T
        xg = np.zeros((B, 2 * (MAX_STEPS + 1)))
        xg[:, 0], xg[:, 1] = X[:, 2 * (dlo - 1)], X[:, 2 * (dlo - 1) + 1]
        sl = slice(TILE * t, TILE * (t + 1))
        for s in range(ns):
            units = [u for u in range(TILE) if (int(msk[s]) >> u) & 1]
            for u in units:               # G_top = Wf^T g_p: raw
                pre[0][:, u] = pre[0][:, u] + xg[:, :2 * MAX_STEPS] @ W0d[u]
            for k in range(1, NL):
                for u in units:
                    tot = pre[k][:, u] + pre[k - 1] @ Wd[k - 1][u]
                    tot = np.where(masks[k - 1][:, sl][:, u], tot, 0.0)
                    pre[k][:, u] = tot if k % 2 == 1 else pre[k - 2][:, u] + tot
            gxm = preF[:, s] + pre[NL - 1] @ WFd[s]
            f = D - 1 - (dlo + s)
            v[:, f], a, b = finish(gxm, f)
            X[:, 2 * (dlo + s)], X[:, 2 * (dlo + s) + 1] = a, b
            xg[:, 2 * (s + 1)], xg[:, 2 * (s + 1) + 1] = a, b
        for l in range(NL):
            S[l][:, sl] = pre[l]
    return (v, S) if return_scratch else v        # S (NL, B, Hp): the published activations in virtual position order
