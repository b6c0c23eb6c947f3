"""Panel-blocked Hermitian tridiagonalisation (LAPACK chetrd / clatrd, lower variant) in numpy.

Groundwork for the next step named in DESIGN.md section 4: the device kernels of this round are the unblocked
algorithm (one rank-2 update of the trailing matrix per reflector, on the vector ALUs); with panels of `nb`
reflectors the trailing update becomes one rank-2 nb product per panel (GEMM-shaped: matrix cores), and the
per-reflector work shrinks to one matrix-vector product with the panel's unmodified trailing matrix plus four
skinny corrections.  This model fixes the algebra in the conventions of tridiag_reg.hip:

  * A is the full Hermitian matrix in the order the kernels see it (arrow row / column first);
  * reflector u has its unit entry at index u + 1 and annihilates column u below it;
    H_u = I - tau_u v_u v_u^H,  v_u[: u + 1] = 0,  v_u[u + 1] = 1;
  * d[u] = T[u, u] (real), e[u] = T[u + 1, u] (real, >= 0 is NOT enforced: LAPACK's sign convention);
  * Q = H_0 H_1 ... H_{n-2},  Q^H A Q = T.

`tests/test_host_logic.py` checks blocked == unblocked (same reflectors, same T) and Q^H A Q = T.
Test infrastructure only: nothing in the product imports this file.
"""
import numpy as np


def _house(x):
    """LAPACK clarfg: x -> (beta, tau, v) with H x = beta e_0, H = I - tau v v^H, v[0] = 1, beta real."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0 and alpha.imag == 0.0:
        return alpha.real, 0.0, np.concatenate([[1.0], np.zeros(len(x) - 1)]).astype(x.dtype)
    beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm ** 2), alpha.real)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return beta, tau, v


def hetrd_unblocked(A):
    """chetd2, lower: returns d[n], e[n-1], V[n, n-1] (column u = v_u), tau[n-1]."""
    A = np.array(A, dtype=np.complex128)
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    V = np.zeros((n, max(n - 1, 0)), dtype=np.complex128)
    tau = np.zeros(max(n - 1, 0), dtype=np.complex128)
    for u in range(n - 1):
        beta, t, v = _house(A[u + 1:, u].copy())
        d[u] = A[u, u].real
        e[u] = beta
        V[u + 1:, u] = v
        tau[u] = t
        if t != 0:
            S = A[u + 1:, u + 1:]
            p = t * (S @ v)
            w = p - 0.5 * t * np.vdot(p, v) * v      # w = p - (tau / 2)(p^H v) v
            S -= np.outer(v, w.conj()) + np.outer(w, v.conj())
    d[n - 1] = A[n - 1, n - 1].real
    return d, e, V, tau


def latrd(A, i0, nb):
    """clatrd, lower: reduce columns i0 .. i0 + nb - 1 of A (trailing matrix NOT updated).
    Returns (d, e, Vp, tau, W) of the panel; Vp, W are [n - i0, nb] in trailing coordinates."""
    n = A.shape[0]
    m = n - i0
    Vp = np.zeros((m, nb), dtype=np.complex128)
    W = np.zeros((m, nb), dtype=np.complex128)
    d = np.zeros(nb)
    e = np.zeros(nb)
    tau = np.zeros(nb, dtype=np.complex128)
    T = A[i0:, i0:]                                  # view: columns are brought up to date one at a time
    for j in range(nb):
        # column j with the updates of the previous reflectors of this panel: a_j -= V W_j^H + W V_j^H
        col = T[j:, j].copy()
        if j > 0:
            col -= Vp[j:, :j] @ W[j, :j].conj() + W[j:, :j] @ Vp[j, :j].conj()
        d[j] = col[0].real
        if j + 1 >= m:
            break
        beta, t, v = _house(col[1:].copy())
        e[j] = beta
        tau[j] = t
        Vp[j + 1:, j] = v
        # w_j = tau (A v - V (W^H v) - W (V^H v)) - (tau / 2)(w^H v) v   on rows j + 1 ..
        S = T[j + 1:, j + 1:]                        # the panel's UNMODIFIED trailing matrix
        p = S @ v
        if j > 0:
            p -= Vp[j + 1:, :j] @ (W[j + 1:, :j].conj().T @ v) + W[j + 1:, :j] @ (Vp[j + 1:, :j].conj().T @ v)
        p *= t
        W[j + 1:, j] = p - 0.5 * t * np.vdot(p, v) * v
    return d, e, Vp, tau, W


def hetrd_blocked(A, nb=16):
    """chetrd, lower, panels of nb: same outputs as hetrd_unblocked."""
    A = np.array(A, dtype=np.complex128)
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    V = np.zeros((n, max(n - 1, 0)), dtype=np.complex128)
    tau = np.zeros(max(n - 1, 0), dtype=np.complex128)
    i0 = 0
    while i0 < n:
        b = min(nb, n - i0)
        dp, ep, Vp, tp, W = latrd(A, i0, b)
        d[i0:i0 + b] = dp
        ne = min(b, n - 1 - i0)
        e[i0:i0 + ne] = ep[:ne]
        tau[i0:i0 + ne] = tp[:ne]
        V[i0:, i0:i0 + ne] = Vp[:, :ne]
        # rank-2 nb update of what lies behind the panel: the GEMM-shaped half of the work
        if i0 + b < n:
            A[i0 + b:, i0 + b:] -= Vp[b:, :] @ W[b:, :].conj().T + W[b:, :] @ Vp[b:, :].conj().T
        i0 += b
    return d, e, V, tau


def form_q(V, tau):
    n = V.shape[0]
    Q = np.eye(n, dtype=np.complex128)
    for u in range(V.shape[1] - 1, -1, -1):
        v = V[:, u]
        Q -= tau[u] * np.outer(v, v.conj() @ Q)
    return Q
