"""Literal CPU restatement of the classical ADMM loop (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/admm.py line by line, INCLUDING the dense inverses and
the broadcasting quirk of the phi update (admm.py:77-79) and the SVD "projection"
that is an identity (admm.py:151-179).  The only substitution is the H step:
the reference calls cvxpy + ECOS (admm.py:140-141, versions unpinned, neither
installed here), we solve the same convex program with scipy SLSQP on its
smooth epigraph form.  PARITY UNPINNED at that boundary: the reference's own
test (admm.py:219-245) prints but records no expected values.
"""
import numpy as np
from scipy.linalg import svd
from scipy.optimize import minimize


def h_step_slsqp(GK_hat, ZK_hat, rho, xbase, ybase, sigma):
    """admm.py:117-148 via a generic solver: min ||h - Re t||^2 s.t. A s + sum h <= 1, |h_i| <= s."""
    n = xbase * ybase
    t = np.real(np.diag(GK_hat + ZK_hat / rho))
    A = float(np.real(2 * np.sqrt(n) * sigma + sigma ** 2))
    if A * np.max(np.abs(t)) + np.sum(t) <= 1.0:
        return np.diag(t.astype(float))
    x0 = np.concatenate([np.zeros(n), [0.0]])
    cons = [{"type": "ineq", "fun": lambda x: 1.0 - A * x[n] - np.sum(x[:n])},
            {"type": "ineq", "fun": lambda x: x[n] - x[:n]},
            {"type": "ineq", "fun": lambda x: x[n] + x[:n]}]
    r = minimize(lambda x: 0.5 * np.sum((x[:n] - t) ** 2), x0, jac=lambda x: np.concatenate([x[:n] - t, [0.0]]),
                 constraints=cons, method="SLSQP", options={"ftol": 1e-15, "maxiter": 1000})
    return np.diag(r.x[:n])


def admm_for_us_literal(y, b, xbase, ybase, lambda_val, sigma, opts=None, use_min_iter=True, min_iter=5,
                        h_step=h_step_slsqp):
    """admm.py:6-114 as written (dense inv, broadcasting add of rho, SVD rebuild)."""
    rho, max_iter, eta_abs, eta_rel = 1.0, 500, 1e-5, 1e-5
    if opts is not None:
        rho = opts.get("rho", rho); max_iter = opts.get("max_iter", max_iter)
        eta_abs = opts.get("eta_abs", eta_abs); eta_rel = opts.get("eta_rel", eta_rel)
    y = y.flatten(); b = b.flatten()
    n = y.shape[0]
    GK = np.zeros((n + 1, n + 1), dtype=complex); ZK = np.zeros((n + 1, n + 1), dtype=complex)
    HK = np.zeros((n, n), dtype=complex)
    it = 0
    for it in range(1, max_iter + 1):
        HK_pre = np.zeros((n, n), dtype=complex) if it == 1 else HK.copy()
        GK_hat = GK[:n, :n]; gK = GK[:n, n]; ZK_hat = ZK[:n, :n]; zetaK = ZK[:n, n]
        diag_inv = np.linalg.inv(np.diag(b * np.conj(b))) + rho * np.ones(n)          # :78 (adds rho everywhere)
        phiK = np.linalg.inv(diag_inv) @ (np.linalg.inv(np.diag(b)) @ y + rho * gK + zetaK)
        HK = h_step(GK_hat, ZK_hat, rho, xbase, ybase, sigma).astype(complex)
        sd = np.zeros((n + 1, n + 1), dtype=complex)
        sd[:n, :n] = HK; sd[:n, n] = phiK; sd[n, :n] = phiK.conj().T; sd[n, n] = 1.0 / (lambda_val ** 2)
        blk = sd.copy()
        sd = sd - ZK / rho
        U, S, Vh = svd(sd)
        S[S < 0] = 0
        Sm = np.zeros_like(sd, dtype=complex); np.fill_diagonal(Sm, S)
        GK = U @ Sm @ Vh
        ZK = ZK + rho * (GK - blk)
        if use_min_iter and it < min_iter:
            continue
        if it > 1:
            eta_pri = eta_abs * np.sqrt(n + 1) + eta_rel * max(np.linalg.norm(GK, "fro"), np.linalg.norm(blk, "fro"))
            eta_dual = eta_abs * np.sqrt(n) + eta_rel * np.linalg.norm(ZK, "fro")
            if np.linalg.norm(GK - blk, "fro") <= eta_pri and np.linalg.norm(rho * (HK - HK_pre), "fro") <= eta_dual:
                break
    return phiK, it


def collapsed_recursion(y, b, rho, iters):
    """with G == sd_Matrix the loop collapses to phi_k = W (y/b + rho phi_{k-1})."""
    n = y.size
    W = np.linalg.inv(np.diag(1.0 / (b * np.conj(b))) + rho * np.ones((n, n)))
    phi = np.zeros(n, dtype=complex)
    for _ in range(iters):
        phi = W @ (y / b + rho * phi)
    return phi
