"""Classical ADMM solver for the delay-Doppler atomic-norm problem (host, complex128).

Mirror of /root/reference/admm.py:6-114 (``admm_for_us``) with the same
signature, option keys, printed messages and return value.  BASELINE.json
config 0 keeps this path on the CPU ("plumbing, no GPU"), so this is host
numpy/scipy code by design -- it is the product's restatement of the
reference's own host function, not a fallback for a GPU path.

Differences from the reference are confined to HOW each step is evaluated:
  * phi step (admm.py:77-79): the reference inverts diag(1/|b|^2) + rho * 1 1^T
    (the ``+ rho*np.ones(len)`` broadcast adds rho to every entry) with three
    dense ``inv``; we apply Sherman-Morrison, O(len).
  * H step (admm.py:117-148): the reference hands the Euclidean projection onto
    {h : (2 sqrt(Nb Nd) sigma + sigma^2) ||h||_inf + sum h <= 1} to cvxpy+ECOS
    (not installed here); we compute that projection exactly (KKT + bisection).
  * G step (admm.py:151-179): kept literally -- scipy SVD, "negative" singular
    values zeroed (there are none), U S Vh rebuilt.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import svd


def project_linf_sum(t: np.ndarray, A: float) -> np.ndarray:
    """argmin_h ||h - t||_2  s.t.  A * max|h_i| + sum h_i <= 1   (A >= 0), t real."""
    t = np.asarray(t, dtype=np.float64)
    if t.size == 0 or A * np.max(np.abs(t)) + np.sum(t) <= 1.0:
        return t.copy()

    def inner(mu):
        u = t - mu
        au = np.sort(np.abs(u))[::-1]
        target = mu * A
        if np.sum(au) <= target:
            s = 0.0
        else:
            # find s >= 0 with sum (|u_i| - s)_+ = target ; piecewise linear, decreasing in s
            cs = np.cumsum(au)
            k = np.arange(1, au.size + 1)
            s_k = (cs - target) / k            # solution if exactly the k largest are clipped
            nxt = np.append(au[1:], 0.0)
            ok = (s_k <= au) & (s_k >= nxt)
            idx = np.argmax(ok) if ok.any() else au.size - 1
            s = max(float(s_k[idx]), 0.0)
        h = np.clip(u, -s, s)
        return h, A * s + np.sum(h) - 1.0

    lo, hi = 0.0, 1.0
    while inner(hi)[1] > 0.0:
        hi *= 2.0
        if hi > 1e300:
            break
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if inner(mid)[1] > 0.0:
            lo = mid
        else:
            hi = mid
    return inner(hi)[0]


def admm_for_us_H_cvx_0(GK_hat, ZK_hat, rho, xbase, ybase, sigma):
    """admm.py:117-148: diagonal real H closest to diag(G + Z/rho) inside the feasible set."""
    Nd, Nb = xbase, ybase
    diag_GZ = np.diag(GK_hat + ZK_hat / rho)
    A = 2 * np.sqrt(Nb * Nd) * sigma + sigma ** 2
    # Frobenius distance to a complex constant has the same minimiser as to its real part
    h = project_linf_sum(np.real(diag_GZ), float(np.real(A)))
    return np.diag(h)


def admm_for_us_G_svd(HK, phiK, lambda_val, ZK, rho):
    """admm.py:151-179 (as written: singular values are never negative, so G == sd_Matrix up to round-off)."""
    len_val = HK.shape[0]
    sd = np.zeros((len_val + 1, len_val + 1), dtype=complex)
    sd[:len_val, :len_val] = HK
    sd[:len_val, len_val] = phiK
    sd[len_val, :len_val] = phiK.conj().T
    sd[len_val, len_val] = 1.0 / (lambda_val ** 2)
    sd = sd - ZK / rho
    U, S, Vh = svd(sd)
    S[S < 0] = 0
    return (U * S) @ Vh


def admm_for_us(y, b, xbase, ybase, lambda_val, sigma, opts=None, use_min_iter=True, min_iter=5):
    """admm.py:6-114.  Returns (phi [len] complex128, iter_count)."""
    rho, max_iter, eta_abs, eta_rel = 1.0, 500, 1e-5, 1e-5
    if opts is not None:
        rho = opts.get("rho", rho)
        max_iter = opts.get("max_iter", max_iter)
        eta_abs = opts.get("eta_abs", eta_abs)
        eta_rel = opts.get("eta_rel", eta_rel)
    y = np.asarray(y).flatten()
    b = np.asarray(b).flatten()
    n = y.shape[0]
    GK = np.zeros((n + 1, n + 1), dtype=complex)
    ZK = np.zeros((n + 1, n + 1), dtype=complex)
    HK = np.zeros((n, n), dtype=complex)
    phiK = np.zeros(n, dtype=complex)
    print(f"Starting ADMM with len_val={n}, max_iter={max_iter}, eta_abs={eta_abs}")
    # (diag(1/|b|^2) + rho 1 1^T)^-1 r  =  d r - d (rho 1^T d r) / (1 + rho 1^T d),  d = |b|^2
    dvec = (b * np.conj(b)).real
    denom = 1.0 + rho * np.sum(dvec)
    iter_count = 0
    for iter_count in range(1, max_iter + 1):
        HK_pre = HK.copy() if iter_count > 1 else np.zeros((n, n), dtype=complex)
        GK_hat, gK = GK[:n, :n], GK[:n, n]
        ZK_hat, zetaK = ZK[:n, :n], ZK[:n, n]
        r = y / b + rho * gK + zetaK
        dr = dvec * r
        phiK = dr - dvec * (rho * np.sum(dr) / denom)
        HK = admm_for_us_H_cvx_0(GK_hat, ZK_hat, rho, xbase, ybase, sigma).astype(complex)
        GK = admm_for_us_G_svd(HK, phiK, lambda_val, ZK, rho)
        blk = np.zeros((n + 1, n + 1), dtype=complex)
        blk[:n, :n] = HK
        blk[:n, n] = phiK
        blk[n, :n] = phiK.conj()
        blk[n, n] = 1.0 / (lambda_val ** 2)
        ZK = ZK + rho * (GK - blk)
        if use_min_iter and iter_count < min_iter:
            continue
        if iter_count > 1:
            eta_pri = eta_abs * np.sqrt(n + 1) + eta_rel * max(np.linalg.norm(GK, "fro"), np.linalg.norm(blk, "fro"))
            eta_dual = eta_abs * np.sqrt(n) + eta_rel * np.linalg.norm(ZK, "fro")
            r_pri = np.linalg.norm(GK - blk, "fro")
            r_dual = np.linalg.norm(rho * (HK - HK_pre), "fro")
            if r_pri <= eta_pri and r_dual <= eta_dual:
                print(f"退出admm迭代，当前迭代次数为: {iter_count}")
                break
    return phiK, iter_count


def cfg1_scene(sig, e, seed=0, snr_w=20.0):
    """The fixed 10 x 10 demo scene of /root/reference/main.py:11-86 with ``data_type = 2`` (QPSK symbols ``sig`` and
    demodulation errors ``e`` taken from data/data.npz, :62-70): three targets, ``Psi = kr(S, conj D) C`` (:15-29),
    ``b = sig - e``, ``y = diag(b + e) Psi + w`` at 20 dB (:73-77), ``sigma = ||e / b|| + 1`` (:81).  Only the noise
    draw is seeded (the reference seeds nothing).  Returns (y [100, 1], b [100], sigma) as main.py passes them."""
    from .synth import steering
    Nb = Nd = 10
    f = np.array([-0.25, 0, 0.14])
    tau = np.array([0.45, 0.25, 0.63])
    C = np.array([-0.5 + 1j, 0.6 - 0.2j, 0.3 + 0.7j])
    S, Dm = steering(f, Nb), steering(tau, Nd)                                   # [L, Nb], [L, Nd]
    Psi = np.einsum("l,li,lj->ij", C, S, np.conj(Dm)).reshape(Nb * Nd, 1)       # kr(S, conj D) @ C
    sig, e = np.asarray(sig).reshape(-1), np.asarray(e).reshape(-1)
    b = sig - e
    real_y = np.diag(b + e) @ Psi
    rng = np.random.default_rng(seed)
    w = np.sqrt(1 / 2) * (rng.standard_normal((Nb * Nd, 1)) + 1j * rng.standard_normal((Nb * Nd, 1)))
    w_var = np.linalg.norm(real_y) ** 2 / (10 ** (snr_w / 10) * Nb * Nd)
    y = real_y + np.sqrt(w_var) * w
    sigma = np.linalg.norm(e / b) + 1
    return y, b, sigma
