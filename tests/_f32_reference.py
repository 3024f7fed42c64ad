"""NumPy float32 restatement of csrc/fsq_fit_f32.h (FSQ_MODE_TEXTBOOK_F32), all fits in lockstep - the "plain fp32 reference of the
same op" the GPU test compares the kernel with.  Same model, start, bounds, pegging, normal equations, Cholesky solve, Nielsen
damping and exit tests; exp / sincos / reciprocals are NumPy's, so individual fits may differ in the last bits and, the fit being
chaotic for a fifth of the ROIs, occasionally by more: the test compares distributions, not bits."""
import numpy as np
f32 = np.float32
LL = np.array([0, 0, 2, 2, .75, .75, 0], f32)
UL = np.array([np.inf, np.inf, 3, 3, 2, 2, 360], f32)
XI = np.repeat(np.arange(5), 5).astype(f32)[None, :]   # first index (row)
YI = np.tile(np.arange(5), 5).astype(f32)[None, :]

def evaluate(x, d):
    """x [n,7] f32, d [n,25] f32 -> chi2 [n], N [n,7,7], g [n,7]"""
    ph = x[:, 6:7] * f32(np.pi / 180)
    cs, sn = np.cos(ph).astype(f32), np.sin(ph).astype(f32)
    A = x[:, 3:4] - XI
    B = x[:, 2:3] - YI
    nu = A * cs - B * sn
    nv = A * sn + B * cs
    i4, i5 = f32(1) / x[:, 4:5], f32(1) / x[:, 5:6]
    u, v = nu * i4, nv * i5
    E = np.exp(f32(-0.5) * (u * u + v * v)).astype(f32)
    g_ = x[:, 0:1] + x[:, 1:2] * E
    r = g_ - d
    aE = x[:, 1:2] * E
    J = np.empty(x.shape[:1] + (25, 7), f32)
    J[:, :, 0] = 1
    J[:, :, 1] = E
    J[:, :, 3] = -aE * (u * cs * i4 + v * sn * i5)
    J[:, :, 2] = -aE * (-u * sn * i4 + v * cs * i5)
    J[:, :, 4] = aE * u * u * i4
    J[:, :, 5] = aE * v * v * i5
    J[:, :, 6] = aE * (u * nv * i4 - v * nu * i5) * f32(np.pi / 180)
    chi2 = np.einsum('ni,ni->n', r, r).astype(f32)
    N = np.einsum('nik,nil->nkl', J, J).astype(f32)
    g = np.einsum('nik,ni->nk', J, r).astype(f32)
    return chi2, N, g

def fit(rois, maxtrips=400, ftol=3e-7, xtol=1e-6, lam0=1e-2, verbose=False):
    d = rois.reshape(-1, 25).astype(f32)
    n = len(d)
    mx, mean, med = d.max(1), d.mean(1, dtype=np.float64).astype(f32), np.median(d, 1).astype(f32)
    ll = np.tile(LL, (n, 1)); ll[:, 1] = (mx - mean) / f32(3)
    ul = np.tile(UL, (n, 1))
    x = np.stack([med, mx, np.full(n, 2.5, f32), np.full(n, 2.5, f32), np.ones(n, f32), np.ones(n, f32), np.zeros(n, f32)], 1)
    x = np.minimum(np.maximum(x, ll), ul)
    chi2, N, g = evaluate(x, d)
    D = np.sqrt(np.einsum('nkk->nk', N)); D[D == 0] = 1
    lam = np.full(n, lam0, f32); nu_ = np.full(n, 2, f32)
    status = np.zeros(n, np.int32); niter = np.ones(n, np.int32); nfev = np.ones(n, np.int32)
    status[chi2 == 0] = 1
    for trip in range(maxtrips):
        act = status == 0
        if not act.any(): break
        peg = ((x <= ll) & (g > 0)) | ((x >= ul) & (g < 0))
        Amat = N + lam[:, None, None] * (D * D)[:, :, None] * np.eye(7, dtype=f32)[None]
        rhs = -g.copy()
        for k in range(7):
            m = peg[:, k]
            Amat[m, k, :] = 0; Amat[m, :, k] = 0; Amat[m, k, k] = 1; rhs[m, k] = 0
        # Cholesky in f32
        Lm = np.zeros_like(Amat); ok = np.ones(n, bool)
        for j in range(7):
            s = Amat[:, j, j] - np.einsum('nk,nk->n', Lm[:, j, :j], Lm[:, j, :j])
            ok &= s > 0
            s = np.where(s > 0, s, 1).astype(f32)
            Lm[:, j, j] = np.sqrt(s)
            for i in range(j + 1, 7):
                Lm[:, i, j] = (Amat[:, i, j] - np.einsum('nk,nk->n', Lm[:, i, :j], Lm[:, j, :j])) / Lm[:, j, j]
        y = np.zeros((n, 7), f32)
        for i in range(7):
            y[:, i] = (rhs[:, i] - np.einsum('nk,nk->n', Lm[:, i, :i], y[:, :i])) / Lm[:, i, i]
        p = np.zeros((n, 7), f32)
        for i in range(6, -1, -1):
            p[:, i] = (y[:, i] - np.einsum('nk,nk->n', Lm[:, i + 1:, i], p[:, i + 1:])) / Lm[:, i, i]
        # step limiting (mpfit.py:1192-1216)
        with np.errstate(divide='ignore', invalid='ignore'):
            tl = np.where((p < 0) & (x + p < ll), (ll - x) / p, np.inf)
            tu = np.where((p > 0) & (x + p > ul), (ul - x) / p, np.inf)
        alpha = np.minimum(1, np.minimum(tl.min(1), tu.min(1))).astype(f32)
        ps = p * alpha[:, None]
        xt = np.minimum(np.maximum(x + ps, ll), ul)
        # snap (mpfit.py:1223-1233)
        xt = np.where(xt >= ul * (1 - f32(1.2e-7)), ul, xt); xt = np.where(xt <= ll * (1 + f32(1.2e-7)), np.where(ll > 0, ll, xt), xt)
        xt = np.minimum(np.maximum(xt, ll), ul).astype(f32)
        chi2t, Nt, gt = evaluate(xt, d)
        pred = -(2 * np.einsum('nk,nk->n', g, ps) + np.einsum('nk,nkl,nl->n', ps, N, ps))
        with np.errstate(divide='ignore', invalid='ignore'):
            rho = np.where(pred > 0, (chi2 - chi2t) / pred, -1)
        acc = act & ok & (rho > 1e-4) & np.isfinite(chi2t)
        rej = act & ~acc
        # termination tests on accepted steps
        actred = np.where(chi2 > 0, (chi2 - chi2t) / chi2, 0)
        prered = np.where(chi2 > 0, pred / chi2, 0)
        dxn = np.sqrt(((D * ps) ** 2).sum(1)); xn = np.sqrt(((D * xt) ** 2).sum(1))
        st = np.zeros(n, np.int32)
        st[acc & (np.abs(actred) <= ftol) & (prered <= ftol)] = 1
        st[acc & (dxn <= xtol * xn) & (st == 0)] = 2
        # accept
        x[acc] = xt[acc]; chi2[acc] = chi2t[acc]; N[acc] = Nt[acc]; g[acc] = gt[acc]
        Dn = np.sqrt(np.einsum('nkk->nk', N)); D = np.where(acc[:, None], np.maximum(D, Dn), D)
        f = np.maximum(1 / 3, 1 - (2 * rho - 1) ** 3)
        lam = np.where(acc, lam * f, lam).astype(f32); nu_ = np.where(acc, 2, nu_).astype(f32)
        lam = np.where(rej, lam * nu_, lam).astype(f32); nu_ = np.where(rej, nu_ * 2, nu_).astype(f32)
        niter += acc; nfev += act
        st[rej & (lam > 1e10)] = 2
        st[act & (st == 0) & (niter >= 200)] = 5
        st[acc & (chi2 == 0)] = 1
        status = np.where(act & (status == 0), st, status)
    status[status == 0] = 5
    return x.astype(np.float64), status, niter, nfev
