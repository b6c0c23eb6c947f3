"""Prototype (CPU, float64 / float32 emulation; not part of the product): the G-layer as a MATRIX FUNCTION instead of an
eigendecomposition, exploiting the structure of the layer matrices -- all but a couple of eigenvalues of A sit in a bulk
of relative width ~1e-4 (measured below on the oracle's own forward), so with the few outliers (lam_k, v_k) deflated,
    E = A - c I - sum_k (lam_k - c) v_k v_k^H,     G = f(A) = f(c) P + f'(c) E + f''(c) E^2 / 2 + sum_k f(lam_k) v_k v_k^H,
P = I - sum_k v_k v_k^H, with a truncation error <= |f'''| ||E||^3 / 6 -- two-vector subspace iteration (converges like
(bulk width / outlier)^iterations) plus ONE Hermitian matrix product on the matrix cores instead of tridiagonalisation + D&C
+ back-transform + rebuild.  f = softplus(lam - thr) * sigmoid(value_net(|lam|)) is smooth away from lam = 0 and the ReLU kinks
of value_net: a matrix whose bulk interval contains one must take the eigensolver.
Prints, per layer of the oracle's forward: bulk width, outliers, whether a kink falls into the bulk, and the error of the
shortcut against the exact eigen-function (relative to max |G|), for default and perturbed weights.
Run:  python tests/proto_spectral_shortcut.py            (evidence for DESIGN.md, "next")"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from admm_net_amd import synth  # noqa: E402
from oracle import admm_net_ref as R  # noqa: E402


def f_and_derivs(sd, k, lam):
    """f, f', f'' at the points lam (float64), by the formulas of admm_net.py:310-334."""
    p = f"gLayers.{k}."
    thr = 1.0 / (1.0 + np.exp(-float(sd[p + "threshold"])))
    w1 = sd[p + "value_net.0.weight"].double().numpy()[:, 0]
    b1 = sd[p + "value_net.0.bias"].double().numpy()
    w2 = sd[p + "value_net.2.weight"].double().numpy()[0]
    b2 = float(sd[p + "value_net.2.bias"])
    x = lam - thr
    s = np.logaddexp(0.0, x)
    s1 = 1.0 / (1.0 + np.exp(-x))
    s2 = s1 * (1.0 - s1)
    a = np.abs(lam)
    sg = np.sign(lam)
    pre = np.outer(a, w1) + b1
    act = pre > 0
    u = (np.maximum(pre, 0.0) * w2).sum(1) + b2
    u1 = (act * (w1 * w2)).sum(1) * sg
    g = 1.0 / (1.0 + np.exp(-u))
    g1 = g * (1 - g) * u1
    g2 = g * (1 - g) * (1 - 2 * g) * u1 * u1
    kinks = np.where(w1 != 0, -b1 / np.where(w1 != 0, w1, 1.0), -1.0)
    return s * g, s1 * g + s * g1, s2 * g + 2 * s1 * g1 + s * g2, kinks[kinks > 0]


def shortcut(A, sd, k, r=2, iters=4, dtype=np.complex128):
    """The matrix-function evaluation; returns (G, info)."""
    n = A.shape[0]
    A = A.astype(dtype)
    rt = np.float64 if dtype == np.complex128 else np.float32
    c0 = rt(np.trace(A).real / n)
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r))).astype(dtype)
    c = c0
    for _ in range(iters):   # shifted subspace iteration; the shift follows the bulk centre as the Ritz values settle (trace / n
        #                      is off by (lam_1 + lam_2) / n, which would set the convergence ratio instead of the bulk width)
        X, _ = np.linalg.qr((A - c * np.eye(n, dtype=dtype)) @ X)
        H = X.conj().T @ (A @ X)
        lam_k, S = np.linalg.eigh((H + H.conj().T) / 2)
        X = (X @ S).astype(dtype)
        c = rt((np.trace(A).real - lam_k.sum()) / (n - r))
    V = X
    E = A - c * np.eye(n, dtype=dtype) - (V * (lam_k - c).astype(rt)) @ V.conj().T
    E2 = E @ E
    delta = float(np.sqrt(np.linalg.norm(E2.astype(np.complex128), "fro")))        # ||E||_2 <= ||E^2||_F^(1/2)
    f0, f1, f2, kinks = f_and_derivs(sd, k, np.array([float(c)]))
    fk, _, _, _ = f_and_derivs(sd, k, lam_k.astype(np.float64))
    P = np.eye(n, dtype=dtype) - V @ V.conj().T
    G = rt(f0[0]) * P + rt(f1[0]) * E + rt(0.5 * f2[0]) * E2 + (V * fk.astype(rt)) @ V.conj().T
    lo, hi = float(c) - delta, float(c) + delta
    bad = (lo <= 0.0 <= hi) or bool(((kinks >= min(abs(lo), abs(hi))) & (kinks <= max(abs(lo), abs(hi)))).any())
    return G, dict(c=float(c), delta=delta, outliers=lam_k, kink_in_bulk=bad)


def main():
    torch.set_num_threads(8)
    for (Nb, Nd, K, perturb, seed) in ((16, 16, 16, 0.0, 0), (16, 16, 16, 0.5, 1), (10, 10, 10, 0.0, 0), (8, 16, 8, 0.5, 2)):
        sd = R.make_weights(Nb, Nd, K, seed=seed, head=False, perturb=perturb)
        y, b, s, _ = synth.make_batch(3, Nb, Nd, seed=20260104)
        tr = []
        R.forward(sd, torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s), Nb, Nd, K, dtype="f64", trace=tr,
                  skip_dead_tail=True)
        print(f"==== grid {Nb}x{Nd}  K={K}  perturb={perturb}")
        for k, t in enumerate(tr):
            A = t["A"].numpy()
            Gex = t["G"].numpy()
            e64, e32, dl, kk = [], [], [], []
            for i in range(A.shape[0]):
                G64, info = shortcut(A[i], sd, k)
                G32, _ = shortcut(A[i], sd, k, dtype=np.complex64)
                sc = np.abs(Gex[i]).max()
                e64.append(np.abs(G64 - Gex[i]).max() / sc)
                e32.append(np.abs(G32.astype(np.complex128) - Gex[i]).max() / sc)
                dl.append(info["delta"])
                kk.append(info["kink_in_bulk"])
            print(f"layer {k:2d}: c {info['c']:8.4f}  ||E|| bound {max(dl):8.2e}  outliers {np.round(info['outliers'], 2)}  "
                  f"kink/zero in bulk {any(kk)!s:5s}  shortcut err f64 {max(e64):8.1e}  fp32 {max(e32):8.1e}")


if __name__ == "__main__":
    main()
