"""Developer diagnostic: eigh residuals of n = 257 matrices at extreme scales (default path vs ADMMNET_* variants)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from admm_net_amd import ops  # noqa: E402
n, rng = 257, np.random.default_rng(11)
X = rng.standard_normal((3, n, n)) + 1j * rng.standard_normal((3, n, n))
A1 = ((X + X.conj().transpose(0, 2, 1)) / 2).astype(np.complex64)
for scale in [1e-17, 1e-14, 1e-10, 1e-5, 1.0, 1e5, 1e10, 1e14, 1e16]:
    As = (A1.astype(np.complex128) * scale).astype(np.complex64)
    w, V = ops.eigh(torch.from_numpy(As).to("cuda:0"))
    w, V = w.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.complex128)
    A64 = As.astype(np.complex128)
    amax = np.abs(A64).max()
    res = np.abs(A64 @ V - V * w[:, None, :]).max() / amax
    orth = np.abs(V.conj().transpose(0, 2, 1) @ V - np.eye(n)).max()
    print("scale %8.1e  residual/amax %9.2e  orth %9.2e  finite %s" % (scale, res, orth, np.isfinite(V).all()))
