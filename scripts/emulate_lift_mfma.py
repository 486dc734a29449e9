"""Lane-level numpy model of `lift_bwd_mfma_kernel` and `lift_runs_mfma_kernel` (st-p3_amd/csrc/stp3_lift.hip).

The matrix-core backward of the voxel pool cannot be run in the build container (no GPU), so its
index arithmetic -- LDS layout, run enumeration and chunking, MFMA operand / result lane maps
(A[l&15][l>>4], B[l>>4][l&15], D[4*(l>>4)+q][l&15], cdna_hip_programming.md section 3) -- is
mirrored here statement by statement and compared with the closed-form gradient of one image
column.  This is a development aid, not a parity test: the GPU test of the real kernel is
tests/test_lift_gpu.py run with STP3_LIFT_BWD=mfma / STP3_LIFT_FWD=mfma.

    python scripts/emulate_lift_mfma.py
"""
import numpy as np

RUN_CAP, ROWS, STRIDE = 128, 32, 66


def mfma_16x16x4(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64][4] per-lane results."""
    A = np.zeros((16, 4), np.float32)
    Bm = np.zeros((4, 16), np.float32)
    lanes = np.arange(64)
    A[lanes & 15, lanes >> 4] = a
    Bm[lanes >> 4, lanes & 15] = b
    Dm = A.astype(np.float64) @ Bm.astype(np.float64)
    out = acc.copy()
    for q in range(4):
        out[:, q] += Dm[4 * (lanes >> 4) + q, lanes & 15].astype(np.float32)
    return out


def kernel_column(fH, D, C, V, feat, prob, vox, gacc, run_cap=RUN_CAP):
    """feat [fH][C], prob [fH][D], vox [fH][D], gacc [V][C] -> grad_feat [fH][C], grad_logits [fH][D]."""
    Dp = D | 1
    fcol = np.zeros(ROWS * STRIDE, np.float32)
    ghat = np.full(run_cap * STRIDE, np.nan, np.float32)      # uninitialised LDS must never matter
    pcol = np.zeros(ROWS * Dp, np.float32)
    tcol = np.full(ROWS * Dp, np.nan, np.float32)
    vcol = np.zeros(ROWS * Dp, np.int64)
    rdesc = np.full(run_cap, -12345, np.int64)
    rvox = np.full(run_cap, -12345, np.int64)
    rcnt = np.zeros(64, np.int64)
    grad_feat = np.full((fH, C), np.nan, np.float32)
    grad_logits = np.full((fH, D), np.nan, np.float32)
    lanes = np.arange(64)

    for wv in range(4):                                        # staging
        for i in range(ROWS // 4):
            h = wv + 4 * i
            row = h < fH
            for lane in range(64):
                fcol[h * STRIDE + lane] = feat[h, lane] if (row and lane < C) else 0.0
                if lane < Dp:
                    ok = row and lane < D
                    pcol[h * Dp + lane] = prob[h, lane] if ok else 0.0
                    vcol[h * Dp + lane] = vox[h, lane] if ok else -1
                    tcol[h * Dp + lane] = 0.0
    for tid in range(64):                                      # run counts
        cnt = 0
        if tid < D:
            prev = -1
            for h in range(fH):
                v = vcol[h * Dp + tid]
                cnt += 1 if (v >= 0 and v != prev) else 0
                prev = v
        rcnt[tid] = cnt
    my_off = np.zeros(256, np.int64)
    total = 0
    for d in range(D):
        c = rcnt[d]
        my_off += np.where(d < np.arange(256), c, 0)
        total += c

    li, kk = lanes & 15, lanes >> 4
    dacc = np.zeros((4, 2, 64, 4), np.float32)                 # [wave][m][lane][q]
    base = 0
    while base < total:
        rc = min(run_cap, total - base)
        rc16 = (rc + 15) & ~15
        for tid in range(D):
            idx, prev, h0 = my_off[tid] - base, -1, 0
            for h in range(fH + 1):
                v = vcol[h * Dp + tid] if h < fH else -1
                if v != prev:
                    if prev >= 0:
                        if 0 <= idx < run_cap:
                            rdesc[idx] = tid | (h0 << 8) | (h << 16)
                            rvox[idx] = prev
                        idx += 1
                    h0, prev = h, v
        for tid in range(256):
            idx = rc + tid
            while idx < rc16:
                rdesc[idx] = 0
                rvox[idx] = -1
                idx += 256
        for wv in range(4):                                    # Ghat
            r0 = wv * 8
            while r0 < rc16:
                for j in range(8):
                    r = r0 + j
                    v = rvox[r] if r < rc16 else -1
                    g = np.where(lanes < C, gacc[max(v, 0), np.minimum(lanes, C - 1)], 0.0) if v >= 0 \
                        else np.zeros(64, np.float32)
                    if r < rc16:
                        ghat[r * STRIDE + lanes] = g
                r0 += 32
        for wv in range(4):                                    # dfeat
            if wv * 16 < C:
                for k0 in range(0, rc16, 4):
                    r = k0 + kk
                    desc = rdesc[r]
                    d, h0, h1 = desc & 255, (desc >> 8) & 255, desc >> 16
                    b = ghat[r * STRIDE + wv * 16 + li]
                    for m in range(2):
                        h = m * 16 + li
                        a = np.where((h >= h0) & (h < h1), pcol[h * Dp + d], 0.0).astype(np.float32)
                        dacc[wv, m] = mfma_16x16x4(a, b, dacc[wv, m])
        for wv in range(4):                                    # dprob
            rt = wv
            while rt * 16 < rc16:
                pacc = np.zeros((2, 64, 4), np.float32)
                for k0 in range(0, 64, 4):
                    b = ghat[(rt * 16 + li) * STRIDE + k0 + kk]
                    for m in range(2):
                        a = fcol[(m * 16 + li) * STRIDE + k0 + kk]
                        pacc[m] = mfma_16x16x4(a, b, pacc[m])
                desc = rdesc[rt * 16 + li]
                d, h0, h1 = desc & 255, (desc >> 8) & 255, desc >> 16
                for m in range(2):
                    for q in range(4):
                        h = m * 16 + kk * 4 + q
                        sel = (h >= h0) & (h < h1)
                        tcol[(h * Dp + d)[sel]] = pacc[m][sel, q]
                rt += 4
        base += run_cap

    for wv in range(4):
        if wv * 16 < C:
            for m in range(2):
                for q in range(4):
                    h = m * 16 + kk * 4 + q
                    sel = h < fH
                    grad_feat[h[sel], (wv * 16 + li)[sel]] = dacc[wv, m][sel, q]
    for wv in range(4):
        for h in range(wv, fH, 4):
            binm = lanes < D
            pr = np.where(binm, pcol[h * Dp + np.minimum(lanes, Dp - 1)], 0.0)
            dp = np.where(binm, tcol[h * Dp + np.minimum(lanes, Dp - 1)], 0.0)
            sdot = np.float32((pr.astype(np.float64) * dp).sum())
            grad_logits[h, :D] = (pr * (dp - sdot))[:D]
    return grad_feat, grad_logits


FWD_ROWS, FWD_STRIDE_F = 32, 80


def fwd_kernel_column(fH, D, C, feat, prob, vox, rb_col, dest_col, n_rows):
    """lift_runs_mfma_kernel for one column.  rb_col [D+1]: exclusive scan of runs per bin (col_base = rb_col[0]);
    dest_col [col_runs]: destination row of every run of the column.  -> runs [n_rows][C] (NaN = never written)."""
    Dp = D | 1
    cap = (fH * Dp + 15) & ~15
    fcol = np.full(FWD_ROWS * FWD_STRIDE_F, np.nan, np.float32)
    pcol = np.full(FWD_ROWS * Dp, np.nan, np.float32)
    vcol = np.full(FWD_ROWS * Dp, -777, np.int64)
    rdesc = np.full(cap, -12345, np.int64)
    rdst = np.full(cap, -12345, np.int64)
    out = np.full((n_rows, C), np.nan, np.float32)
    lanes = np.arange(64)
    for wv in range(4):
        for i in range(FWD_ROWS // 4):
            h = wv + 4 * i
            row = h < fH
            for lane in range(64):
                fcol[h * FWD_STRIDE_F + lane] = feat[h, lane] if (row and lane < C) else 0.0
                if lane < Dp:
                    ok = row and lane < D
                    pcol[h * Dp + lane] = prob[h, lane] if ok else 0.0
                    vcol[h * Dp + lane] = vox[h, lane] if ok else -1
    col_base, col_runs = rb_col[0], rb_col[D] - rb_col[0]
    runs16 = (col_runs + 15) & ~15
    for tid in range(256):
        i = tid
        while i < runs16:
            rdst[i] = dest_col[i] if i < col_runs else -1
            if i >= col_runs:
                rdesc[i] = 0
            i += 256
    for tid in range(D):
        idx, prev, h0 = rb_col[tid] - col_base, -1, 0
        for h in range(fH + 1):
            v = vcol[h * Dp + tid] if h < fH else -1
            if v != prev:
                if prev >= 0:
                    rdesc[idx] = tid | (h0 << 8) | (h << 16)
                    idx += 1
                h0, prev = h, v
    li, kk = lanes & 15, lanes >> 4
    for wv in range(4):
        if wv * 16 >= C:
            continue
        bq = [fcol[(4 * ks + kk) * FWD_STRIDE_F + wv * 16 + li] for ks in range(FWD_ROWS // 4)]
        for r0 in range(0, runs16, 16):
            desc = rdesc[r0 + li]
            d, h0, h1 = desc & 255, (desc >> 8) & 255, desc >> 16
            acc = np.zeros((64, 4), np.float32)
            for ks in range(FWD_ROWS // 4):
                h = 4 * ks + kk
                a = np.where((h >= h0) & (h < h1), pcol[h * Dp + d], 0.0).astype(np.float32)
                acc = mfma_16x16x4(a, bq[ks], acc)
            for q in range(4):
                row = rdst[r0 + 4 * kk + q]
                sel = row >= 0
                out[row[sel], (wv * 16 + li)[sel]] = acc[sel, q]
    return out


def fwd_reference(fH, D, C, feat, prob, vox, rb_col, dest_col, n_rows):
    out = np.full((n_rows, C), np.nan, np.float64)
    for d in range(D):
        j, h = 0, 0
        while h < fH:
            v = vox[h, d]
            h1 = h
            while h1 < fH and vox[h1, d] == v:
                h1 += 1
            if v >= 0:
                row = dest_col[rb_col[d] - rb_col[0] + j]
                out[row] = (prob[h:h1, d].astype(np.float64)[:, None] * feat[h:h1].astype(np.float64)).sum(0)
                j += 1
            h = h1
        assert j == rb_col[d + 1] - rb_col[d]
    return out


def closed_form(feat, prob, vox, gacc):
    fH, D = prob.shape
    g = np.where((vox >= 0)[..., None], gacc[np.maximum(vox, 0)], 0.0).astype(np.float64)   # [fH][D][C]
    dprob = np.einsum('hdc,hc->hd', g, feat.astype(np.float64))
    dfeat = np.einsum('hdc,hd->hc', g, prob.astype(np.float64))
    dlogit = prob * (dprob - (prob * dprob).sum(1, keepdims=True))
    return dfeat, dlogit


def random_column(rng, fH, D, C, V, mean_run, p_invalid):
    vox = np.zeros((fH, D), np.int64)
    for d in range(D):
        h = 0
        while h < fH:
            n = 1 + rng.geometric(1.0 / mean_run) - 1
            v = -1 if rng.random() < p_invalid else rng.integers(0, V)
            vox[h:h + n, d] = v
            h += n
    feat = rng.standard_normal((fH, C)).astype(np.float32)
    logits = rng.standard_normal((fH, D)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    gacc = rng.standard_normal((V, C)).astype(np.float32)
    return feat, prob, vox, gacc


def main():
    rng = np.random.default_rng(0)
    cases = [
        # fH, D, C, V, mean run, p(invalid), run cap
        (28, 48, 64, 500, 3.0, 0.2, RUN_CAP),     # the bench geometry, several chunks
        (28, 48, 64, 500, 12.0, 0.5, RUN_CAP),    # long runs, one chunk, not a multiple of 16
        (32, 41, 32, 50, 1.0, 0.0, RUN_CAP),      # every point its own run, C = 32, full 32 rows
        (5, 3, 16, 7, 2.0, 0.9, RUN_CAP),         # tiny, mostly invalid
        (28, 48, 64, 500, 2.0, 1.0, RUN_CAP),     # nothing valid at all
        (17, 64, 48, 99, 2.5, 0.1, 32),           # D = 64, small chunk capacity
    ]
    worst = 0.0
    for fH, D, C, V, mr, pi, cap in cases:
        feat, prob, vox, gacc = random_column(rng, fH, D, C, V, mr, pi)
        gf, gl = kernel_column(fH, D, C, V, feat, prob, vox, gacc, run_cap=cap)
        rf, rl = closed_form(feat, prob, vox, gacc)
        assert np.isfinite(gf).all() and np.isfinite(gl).all(), 'uninitialised output'
        ef = np.abs(gf - rf).max() / max(np.abs(rf).max(), 1e-6)
        el = np.abs(gl - rl).max() / max(np.abs(rl).max(), 1e-6)
        worst = max(worst, ef, el)
        print(f'fH={fH} D={D} C={C} runs/col={int((np.diff(vox, axis=0, prepend=-7) != 0).sum())}: '
              f'dfeat rel err {ef:.2e}, dlogit rel err {el:.2e}')
        assert ef < 1e-5 and el < 1e-5
    print('backward OK, worst', worst)
    # forward: stage 1 as a GEMM per column
    worst = 0.0
    for fH, D, C, V, mr, pi, _ in cases:
        feat, prob, vox, _ = random_column(rng, fH, D, C, V, mr, pi)
        starts = (np.diff(vox, axis=0, prepend=-7) != 0) & (vox >= 0)
        per_bin = starts.sum(0)
        rb_col = 1000 + np.concatenate([[0], np.cumsum(per_bin)])          # col_base = 1000: ids are plan-global
        col_runs = int(per_bin.sum())
        n_rows = col_runs + 9
        dest_col = rng.permutation(n_rows)[:col_runs]
        got = fwd_kernel_column(fH, D, C, feat, prob, vox, rb_col, dest_col, n_rows)
        ref = fwd_reference(fH, D, C, feat, prob, vox, rb_col, dest_col, n_rows)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), 'rows written / left alone differ'
        m = ~np.isnan(ref)
        err = np.abs(got[m] - ref[m]).max() / max(np.abs(ref[m]).max(), 1e-6) if m.any() else 0.0
        worst = max(worst, err)
        print(f'fwd fH={fH} D={D} C={C} runs/col={col_runs}: rel err {err:.2e}')
        assert err < 1e-5
    print('forward OK, worst', worst)


if __name__ == '__main__':
    main()
