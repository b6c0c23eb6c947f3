"""Developer diagnostic: accuracy of the T factors and of the block-reflector back-transform in isolation.
Runs admmnet_eigh_c64 on one 257 x 257 matrix with a caller-held workspace, then reads the intermediate images
(reflector rows, taus, T factors, D&C eigenvectors W, final V^T) out of it and redoes each stage in float64."""
import ctypes
import os
import sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from admm_net_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n, D = 257, 256
rng = np.random.default_rng(int(os.environ.get("WY_SEED", "0")))
U = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
A = 0.3 * np.eye(n) + np.diag(0.05 * rng.standard_normal(n)) + U @ np.diag([5.0, -7.0, 2.0]) @ U.conj().T
A = ((A + A.conj().T) / 2).astype(np.complex64)
tA = torch.from_numpy(A[None]).to(dev)
need = lib.admmnet_eigh_workspace_bytes(n, 1)
ws = torch.zeros(need, dtype=torch.uint8, device=dev)
w = torch.empty(1, n, dtype=torch.float32, device=dev)
V = torch.empty(1, n, n, dtype=torch.complex64, device=dev)
st = torch.zeros(4, dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
_lib.check(lib.admmnet_eigh_c64(n, 1, p(tA), p(w), p(V), p(ws), need, p(st), ctypes.c_void_p(0)), "eigh")
torch.cuda.synchronize()
raw = ws.cpu().numpy()
al = lambda v: (v + 255) // 256 * 256
off = 0
def take(nbytes):
    global off
    o = off
    off = al(off + nbytes)
    return o
o_M = take(8 * (D * D + D + 1)); o_QV = take(4 * n * 2 * D); o_dT = take(4 * n * 64); o_eT = take(4 * n * 64)
o_w = take(4 * n); o_w0 = take(4 * n); o_logn = take(8); o_W = take(4 * 3 * n * n); o_VT = take(4 * n * 2 * D); o_T = take(8 * 17 * 256)
assert off == need, (off, need)
Mg = raw[o_M:o_M + 8 * (D * D + D + 1)].view(np.complex64)
refl = Mg[:D * D].reshape(D, D).astype(np.complex128)          # row u = v_u
taus = Mg[D * D:D * D + D].astype(np.complex128)
Tg = raw[o_T:o_T + 8 * 17 * 256].view(np.complex64).reshape(17, 16, 16).astype(np.complex128)
VT = raw[o_VT:o_VT + 4 * n * 2 * D].view(np.float32).reshape(n, 2 * D).astype(np.float64)
Vgpu = (VT[:, :D] + 1j * VT[:, D:]).T                           # [rho][c]
Wall = raw[o_W:o_W + 4 * 3 * n * n].view(np.float32).reshape(3, n, n).astype(np.float64)
# which ping-pong buffer holds the final WT: the one whose back-transform matches the GPU's V (the other holds the
# previous merge level, orthogonal as well)
def back(WTc):
    Xc = WTc.T[1:, :].astype(np.complex128)
    for u in range(D - 1, -1, -1):
        v = refl[u]
        Xc = Xc - np.outer(taus[u] * v, v.conj() @ Xc)
    return Xc
WT = min((Wall[i] for i in range(2)), key=lambda Xc: np.abs(Vgpu - back(Xc)).max())
X0 = WT.T[1:, :]                                                # W[1:, :]
# (1) T factors vs float64 recomputation from the same reflectors / taus
worst = 0.0
for pp in range(17):
    u0 = 16 * (pp - 1) + 1
    T = np.zeros((16, 16), np.complex128)
    Y = np.zeros((D, 16), np.complex128)
    g = np.zeros(16, np.complex128)
    for i in range(16):
        u = u0 + i
        if 0 <= u < D:
            Y[:, i] = refl[u]
            g[i] = taus[u]
    for i in range(16):
        T[i, i] = g[i]
        if i:
            T[:i, i] = -g[i] * (T[:i, :i] @ (Y[:, :i].conj().T @ Y[:, i]))
    sc = np.abs(T).max() + 1e-300
    e = np.abs(T - Tg[pp]).max() / sc
    worst = max(worst, e)
    if e > 1e-5:
        print(f"panel {pp}: T rel err {e:.2e}")
print(f"T factors: worst relative error {worst:.2e}")
# (2) the back-transform alone: V = Q' X0 in float64 from the same reflectors vs the GPU image
X = X0.astype(np.complex128)
for u in range(D - 1, -1, -1):
    v = refl[u]
    X = X - np.outer(taus[u] * v, v.conj() @ X)
print(f"back-transform: max|V_gpu - V_f64| {np.abs(Vgpu - X).max():.2e}   (max|V| {np.abs(X).max():.2e})")
Vfull = np.vstack([WT.T[0:1, :].astype(np.complex128), Vgpu])
print(f"orth of the GPU V: {np.abs(Vfull.conj().T @ Vfull - np.eye(n)).max():.2e};  of the f64 back-transform: "
      f"{np.abs(np.vstack([WT.T[0:1, :], X]).conj().T @ np.vstack([WT.T[0:1, :], X]) - np.eye(n)).max():.2e}")
