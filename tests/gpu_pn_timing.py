"""Developer diagnostic: phase cycle counts of the panel tridiagonalisation (ADMMNET_PN_TIMING=1) on random matrices."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from admm_net_amd import ops  # noqa: E402
B, n = int(os.environ.get("PN_B", "512")), 257
rng = np.random.default_rng(0)
X = rng.standard_normal((B, n, n)).astype(np.float32) + 1j * rng.standard_normal((B, n, n)).astype(np.float32)
A = ((X + X.conj().transpose(0, 2, 1)) / 2).astype(np.complex64)
w, V = ops.eigh(torch.from_numpy(A).to("cuda:0"))
torch.cuda.synchronize()
w = w.cpu().numpy().astype(np.float64)
ref = np.linalg.eigvalsh(A[:4].astype(np.complex128))
print("eig err", np.abs(np.sort(w[:4], 1) - ref).max() / np.abs(ref).max())
