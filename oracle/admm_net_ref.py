"""CPU restatement of the unrolled ADMM-Net forward (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/admm_net.py function by function (file:line cited on
each function) but is written functionally over a plain ``dict`` of weights
that uses the reference's ``state_dict`` key names.  ``dtype='f32'`` mirrors
the reference arithmetic (float32 / complex64); ``dtype='f64'`` evaluates the
same formulas in float64 / complex128 and is the ground truth used to show the
HIP path is as accurate as the reference itself.

Pinned against the imported reference by tests/golden/make_golden.py; the
fixtures it wrote live in tests/golden/*.npz and are re-checked on every run of
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

EPS = 1e-8  # admm_net.py:74,114,211,360 (every layer's epsilon)


def _dt(dtype: str):
    if dtype == "f32":
        return torch.float32, torch.complex64
    if dtype == "f64":
        return torch.float64, torch.complex128
    raise ValueError(dtype)


def cast_weights(sd: Dict[str, torch.Tensor], dtype: str) -> Dict[str, torch.Tensor]:
    rt, _ = _dt(dtype)
    return {k: v.detach().to(rt) for k, v in sd.items()}


# --------------------------------------------------------------------------
# per-layer pieces
# --------------------------------------------------------------------------
def phi_layer(sd, k, y, b, G, Z):
    """admm_net.py:79-105 (PhiLayer.forward)."""
    g = G[:, :-1, -1]
    zeta = Z[:, :-1, -1]
    b_sq = torch.abs(b) ** 2 + EPS
    rho = F.softplus(sd[f"phiLayers.{k}.rho"])
    weight = b_sq / (1 + rho * b_sq)
    y_over_b = y / (b + EPS)
    return weight * (y_over_b + rho * g + zeta)


def h_layer(sd, k, G, Z, sigma, M, N):
    """admm_net.py:134-194 (HLayer.forward + _differentiable_projection).

    Returns the diagonal ``h`` [B, D]; the reference materialises diag_embed(h).
    """
    D = M * N
    p = f"hLayers.{k}."
    rho = F.softplus(sd[p + "rho"])
    T = G[:, :D, :D] + Z[:, :D, :D] / (rho + EPS)
    t = torch.diagonal(T, dim1=1, dim2=2).real
    sigma = sigma.to(t.dtype)
    A = 2 * torch.sqrt(torch.tensor(float(M * N), dtype=torch.float32)).to(t.dtype) * sigma + sigma ** 2
    A = A.reshape(-1, 1)
    hid = F.relu(F.linear(t, sd[p + "correction_net.0.weight"], sd[p + "correction_net.0.bias"]))
    corr = torch.tanh(F.linear(hid, sd[p + "correction_net.2.weight"], sd[p + "correction_net.2.bias"]))
    tc = t + 0.1 * corr
    linf = torch.max(torch.abs(tc), dim=1, keepdim=True)[0]
    tr = torch.sum(tc, dim=1, keepdim=True)
    cval = A * linf + tr
    scale = torch.sigmoid(sd[p + "projection_weight"]) / (cval + EPS)
    scale = torch.clamp(scale, max=1.0)
    return tc * scale


def block_matrix(phi, h, corner: float):
    """[[diag(h), phi], [phi^H, corner]] -- admm_net.py:273-284 / 428-439."""
    B, D = phi.shape
    n = D + 1
    C = torch.zeros(B, n, n, dtype=phi.dtype)
    idx = torch.arange(D)
    C[:, idx, idx] = h.to(phi.dtype)
    C[:, :D, D] = phi
    C[:, D, :D] = phi.conj()
    C[:, D, D] = corner
    return C


def eigenvalue_map(sd, k, w):
    """admm_net.py:310-334 (GLayer._eigenvalues_projection), vectorised."""
    p = f"gLayers.{k}."
    thr = torch.sigmoid(sd[p + "threshold"])
    base = F.softplus(w - thr)
    a = w.abs().unsqueeze(-1)
    hid = F.relu(F.linear(a, sd[p + "value_net.0.weight"], sd[p + "value_net.0.bias"]))
    sc = torch.sigmoid(F.linear(hid, sd[p + "value_net.2.weight"], sd[p + "value_net.2.bias"]))
    return base * sc.squeeze(-1)


def g_layer(sd, k, phi, h, Z, return_eig=False):
    """admm_net.py:237-354 (GLayer.forward)."""
    p = f"gLayers.{k}."
    lam = F.softplus(sd[p + "lambda_param"])
    corner = float(1.0 / (lam ** 2 + EPS))          # .item() at :271
    C = block_matrix(phi, h, corner)
    rho = F.softplus(sd[p + "rho"])
    A = C - (1.0 / (rho + EPS)) * Z                  # :288
    Ah = 0.5 * (A + A.transpose(1, 2).conj())        # :301
    w, V = torch.linalg.eigh(Ah)                     # :303
    wp = eigenvalue_map(sd, k, w)
    Gn = torch.bmm(V, torch.bmm(torch.diag_embed(wp.to(V.dtype)), V.transpose(1, 2).conj()))
    Gn = 0.5 * (Gn + Gn.transpose(1, 2).conj())      # :352
    if return_eig:
        return Gn, w, Ah
    return Gn


def z_step(sd, k, res_norm, mean_norm=None):
    """admm_net.py:443-474 (ZLayer._compute_adaptive_step) given r_b = ||R_b||_F.

    ``mean_norm`` overrides the batch mean (multi-GPU global scope).
    Returns adaptive_rho [B].
    """
    p = f"zLayers.{k}."
    rho = F.softplus(sd[p + "rho"])
    B = res_norm.shape[0]
    rt = res_norm.dtype
    k_norm = torch.tensor(k / 10.0, dtype=torch.float32).to(rt).repeat(B)   # :457 (fp32 tensor)
    rho_norm = torch.full((B,), float(rho), dtype=rt)                        # :458 (.item())
    mean = res_norm.mean() if mean_norm is None else mean_norm
    rn = res_norm / (mean + EPS)
    feat = torch.stack([k_norm, rho_norm, rn], dim=1)
    hid = F.relu(F.linear(feat, sd[p + "residual_scale_net.0.weight"], sd[p + "residual_scale_net.0.bias"]))
    sf = torch.sigmoid(F.linear(hid, sd[p + "residual_scale_net.2.weight"], sd[p + "residual_scale_net.2.bias"]))
    sf = 0.5 + 1.5 * sf
    return rho * sf.squeeze(1)


def z_layer(sd, k, phi, h, G, Z, mean_norm=None, return_aux=False):
    """admm_net.py:388-414 (ZLayer.forward)."""
    p = f"zLayers.{k}."
    lam = F.softplus(sd[p + "lambda_param"])
    corner = float(1.0 / (lam ** 2 + EPS))          # :425-426
    Cz = block_matrix(phi, h, corner)
    R = G - Cz
    rn = torch.linalg.norm(R, dim=(1, 2))            # :454 Frobenius
    arho = z_step(sd, k, rn, mean_norm)
    Zn = Z + arho.reshape(-1, 1, 1) * R
    if return_aux:
        return Zn, rn, arho
    return Zn


def peak_head(sd, phi, M, N, L=3, hidden=128, heads=4):
    """admm_net.py:570-630 (PeakSearchLayer.forward), eval mode (no dropout)."""
    p = "peakSearchLayer."
    B = phi.shape[0]
    feat = torch.cat([phi.real, phi.imag], dim=1)
    x = F.relu(F.linear(feat, sd[p + "feature_extractor.0.weight"], sd[p + "feature_extractor.0.bias"]))
    x = F.relu(F.linear(x, sd[p + "feature_extractor.2.weight"], sd[p + "feature_extractor.2.bias"]))
    pos = F.linear(sd[p + "position_encoder"], sd[p + "position_projection.weight"],
                   sd[p + "position_projection.bias"])                       # [D, hidden]
    Wi, bi = sd[p + "attention.in_proj_weight"], sd[p + "attention.in_proj_bias"]
    q = F.linear(x, Wi[:hidden], bi[:hidden])                                # [B, hidden]
    kk = F.linear(pos, Wi[hidden:2 * hidden], bi[hidden:2 * hidden])         # [D, hidden]
    vv = F.linear(pos, Wi[2 * hidden:], bi[2 * hidden:])
    hd = hidden // heads
    qh = q.reshape(B, heads, hd)
    kh = kk.reshape(-1, heads, hd)
    vh = vv.reshape(-1, heads, hd)
    sc = torch.einsum("bhd,thd->bht", qh, kh) / math.sqrt(hd)
    at = torch.softmax(sc, dim=-1)
    ctx = torch.einsum("bht,thd->bhd", at, vh).reshape(B, hidden)
    att = F.linear(ctx, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
    xf = x + att
    xp = xf
    for i in (0, 2, 4):
        xp = F.relu(F.linear(xp, sd[p + f"peak_extractor.{i}.weight"], sd[p + f"peak_extractor.{i}.bias"]))
    taus, fs, cs = [], [], []
    for t in range(L):
        tf = xp + torch.tensor(t / L, dtype=torch.float32).to(xp.dtype)     # :615 fp32 tensor
        a = F.relu(F.linear(tf, sd[p + f"tau_regressor.{t}.0.weight"], sd[p + f"tau_regressor.{t}.0.bias"]))
        taus.append(torch.sigmoid(F.linear(a, sd[p + f"tau_regressor.{t}.2.weight"], sd[p + f"tau_regressor.{t}.2.bias"])))
        a = F.relu(F.linear(tf, sd[p + f"f_regressor.{t}.0.weight"], sd[p + f"f_regressor.{t}.0.bias"]))
        fs.append(torch.tanh(F.linear(a, sd[p + f"f_regressor.{t}.2.weight"], sd[p + f"f_regressor.{t}.2.bias"])))
        a = F.relu(F.linear(tf, sd[p + "confidence_net.0.weight"], sd[p + "confidence_net.0.bias"]))
        cs.append(torch.sigmoid(F.linear(a, sd[p + "confidence_net.2.weight"], sd[p + "confidence_net.2.bias"])))
    return torch.cat(taus, 1), torch.cat(fs, 1), torch.cat(cs, 1)


# --------------------------------------------------------------------------
# whole forward
# --------------------------------------------------------------------------
@torch.no_grad()
def forward(sd, y, b, sigma, M, N, K, L=3, dtype="f32", head=False,
            trace: Optional[List[dict]] = None, skip_dead_tail=False,
            mean_norm_fn=None):
    """admm_net.py:742-764 (PhiEstADMMNet.forward) / :791-816 (ADMMNet.forward).

    ``trace``: list that receives per-layer dicts {phi,h,w,G,Z,rn,arho}.
    ``skip_dead_tail``: do not run H/G/Z of the last layer (its outputs are
    never used, admm_net.py:757-764); the returned phi is identical.
    ``mean_norm_fn(k, rn) -> scalar`` lets the multi-rank tests inject the
    global batch mean.
    """
    rt, ct = _dt(dtype)
    sd = cast_weights(sd, dtype)
    y = y.to(ct)
    b = b.to(ct)
    sigma = sigma.to(rt).reshape(-1)
    B = y.shape[0]
    D = M * N
    n = D + 1
    G = torch.zeros(B, n, n, dtype=ct)   # reference: float32 zeros, promoted after layer 0 (values identical)
    Z = torch.zeros(B, n, n, dtype=ct)
    phi = None
    for k in range(K):
        phi = phi_layer(sd, k, y, b, G, Z)
        if skip_dead_tail and k == K - 1:
            break
        h = h_layer(sd, k, G, Z, sigma, M, N)
        G, w, A = g_layer(sd, k, phi, h, Z, return_eig=True)
        mn = None
        if mean_norm_fn is not None:
            Cz_corner = float(1.0 / (F.softplus(sd[f"zLayers.{k}.lambda_param"]) ** 2 + EPS))
            rn_loc = torch.linalg.norm(G - block_matrix(phi, h, Cz_corner), dim=(1, 2))
            mn = mean_norm_fn(k, rn_loc)
        Z, rn, arho = z_layer(sd, k, phi, h, G, Z, mean_norm=mn, return_aux=True)
        if trace is not None:
            trace.append(dict(phi=phi.clone(), h=h.clone(), w=w.clone(), A=A.clone(),
                              G=G.clone(), Z=Z.clone(), rn=rn.clone(), arho=arho.clone()))
    if head:
        tau, f, conf = peak_head(sd, phi, M, N, L)
        return tau, f, conf, phi
    return phi


# --------------------------------------------------------------------------
# weights with the reference's key set (admm_net.py:76,121-132,218-235,368-386,502-555)
# --------------------------------------------------------------------------
def _linear(gen, out_f, in_f, prefix, sd):
    bound = 1.0 / math.sqrt(in_f)
    sd[prefix + "weight"] = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    sd[prefix + "bias"] = (torch.rand(out_f, generator=gen) * 2 - 1) * bound


def make_weights(M, N, K, L=3, seed=0, head=True, perturb=0.0, hidden=128) -> Dict[str, torch.Tensor]:
    """Random weights carrying exactly the reference state_dict key set.

    ``perturb`` > 0 draws every scalar parameter from N(init, perturb) to break
    the "everything = softplus(1)" degeneracy of the default init.
    """
    gen = torch.Generator().manual_seed(seed)
    D = M * N
    sd: Dict[str, torch.Tensor] = {}

    def scalar(name, init):
        v = torch.tensor(float(init))
        if perturb > 0:
            v = v + perturb * torch.randn((), generator=gen)
        sd[name] = v

    for k in range(K):
        scalar(f"phiLayers.{k}.rho", 1.0)
        scalar(f"hLayers.{k}.rho", 1.0)
        scalar(f"hLayers.{k}.projection_weight", 1.0)
        _linear(gen, 64, D, f"hLayers.{k}.correction_net.0.", sd)
        _linear(gen, D, 64, f"hLayers.{k}.correction_net.2.", sd)
        scalar(f"gLayers.{k}.lambda_param", 0.1)
        scalar(f"gLayers.{k}.rho", 1.0)
        scalar(f"gLayers.{k}.threshold", 0.0)
        _linear(gen, 16, 1, f"gLayers.{k}.value_net.0.", sd)
        _linear(gen, 1, 16, f"gLayers.{k}.value_net.2.", sd)
        scalar(f"zLayers.{k}.rho", 1.0)
        scalar(f"zLayers.{k}.lambda_param", 1.0)
        _linear(gen, 32, 3, f"zLayers.{k}.residual_scale_net.0.", sd)
        _linear(gen, 1, 32, f"zLayers.{k}.residual_scale_net.2.", sd)
        _linear(gen, 8, 3, f"zLayers.{k}.step_adjust_net.0.", sd)    # unused by forward, present in state_dict
        _linear(gen, 1, 8, f"zLayers.{k}.step_adjust_net.2.", sd)
    if head:
        p = "peakSearchLayer."
        tg, fg = torch.meshgrid(torch.linspace(0, 1, M), torch.linspace(-0.5, 0.5, N), indexing="ij")
        sd[p + "position_encoder"] = torch.stack([tg.flatten(), fg.flatten()], dim=1)
        _linear(gen, hidden, 2 * D, p + "feature_extractor.0.", sd)
        _linear(gen, hidden, hidden, p + "feature_extractor.2.", sd)
        _linear(gen, hidden, 2, p + "position_projection.", sd)
        bnd = math.sqrt(6.0 / (3 * hidden + hidden))
        sd[p + "attention.in_proj_weight"] = (torch.rand(3 * hidden, hidden, generator=gen) * 2 - 1) * bnd
        sd[p + "attention.in_proj_bias"] = 0.1 * torch.randn(3 * hidden, generator=gen) if perturb > 0 else torch.zeros(3 * hidden)
        _linear(gen, hidden, hidden, p + "attention.out_proj.", sd)
        if perturb == 0:
            sd[p + "attention.out_proj.bias"] = torch.zeros(hidden)
        _linear(gen, hidden // 2, hidden, p + "peak_extractor.0.", sd)
        _linear(gen, hidden // 4, hidden // 2, p + "peak_extractor.2.", sd)
        _linear(gen, hidden // 8, hidden // 4, p + "peak_extractor.4.", sd)
        for t in range(L):
            _linear(gen, 32, hidden // 8, p + f"tau_regressor.{t}.0.", sd)
            _linear(gen, 1, 32, p + f"tau_regressor.{t}.2.", sd)
            _linear(gen, 32, hidden // 8, p + f"f_regressor.{t}.0.", sd)
            _linear(gen, 1, 32, p + f"f_regressor.{t}.2.", sd)
        _linear(gen, 16, hidden // 8, p + "confidence_net.0.", sd)
        _linear(gen, 1, 16, p + "confidence_net.2.", sd)
    return sd


def flops_per_signal(K, n, D, natoms=0):
    """F = (K-1)*24 n^3 + K*256 D + 8 D Natoms."""
    return (K - 1) * 24.0 * n ** 3 + K * 256.0 * D + 8.0 * D * natoms
