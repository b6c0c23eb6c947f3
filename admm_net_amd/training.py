"""Differentiable forward of the unrolled network for the reference's training scripts
(``trainPhi.py`` / ``train.py``).

The inference path (``modules._FusedBase._run``) keeps nothing a backward pass could use, so training
takes this route instead: the layers are evaluated as differentiable tensor operations on the GPU and
the Hermitian eigendecomposition -- the dominant cost of every layer (admm_net.py:303) -- runs on the
HIP eigensolver behind ``admmnet_eigh_c64`` with the gradient the reference defines for it: the
eigenvectors are detached (admm_net.py:306), so only the eigenvalues carry gradient,
``dL/dA = V diag(dL/dw) V^H``.  The two n^3 contractions around it -- the rebuild ``G = V f(L) V^H`` of
admm_net.py:336-354 (forward, and its adjoint ``Re(v_c^H S v_c)`` in the backward) and that ``V diag(gw) V^H`` -- run on
the library's matrix-core kernels too (``admmnet_vdvh_c64`` / ``admmnet_vhsv_f32``, csrc/vdvh.hip), not on framework
GEMMs; the remaining layer steps are O(n^2) tensor operations.

Gradient flow mirrors the reference as written:
  * the corner values ``1 / (softplus(lambda)^2 + eps)`` go through ``.item()`` (admm_net.py:271, 426):
    ``gLayers.k.lambda_param`` / ``zLayers.k.lambda_param`` receive no gradient;
  * the ``rho`` FEATURE of the step network is a detached number (admm_net.py:458), the multiplying
    ``rho_base`` is not (admm_net.py:469);
  * the residual norm is divided by the mean over the batch the call sees (admm_net.py:459);
  * H, G, Z of the last layer are dead (admm_net.py:757-764): they are not evaluated, and, as in the
    reference, their parameters end up with ``grad = None``.
There is no CPU fallback: without the HIP library ``ops.eigh`` raises.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import ops

EPS = 1e-8   # every layer's epsilon (admm_net.py:74, 114, 211, 360)


class Assembler:
    """The two contractions with a constant eigenvector matrix: ``vdvh(V, d) = V diag(d) V^H`` (exactly Hermitian)
    and its adjoint ``vhsv(V, S)[c] = Re(v_c^H S v_c)`` for Hermitian S.  The product uses the HIP kernels; the CPU unit
    tests hand in ``TorchAssembler`` to check the autograd wiring without a GPU."""

    vdvh = staticmethod(ops.vdvh)
    vhsv = staticmethod(ops.vhsv)


class TorchAssembler:
    """Stand-in for tests (tensor operations on whatever device the inputs live on)."""

    @staticmethod
    def vdvh(V, d):
        G = torch.matmul(V * d.unsqueeze(1).to(V.dtype), V.transpose(1, 2).conj())
        return 0.5 * (G + G.transpose(1, 2).conj())

    @staticmethod
    def vhsv(V, S):
        return (V.conj() * torch.matmul(S, V)).sum(dim=1).real


class _EighValuesOnly(torch.autograd.Function):
    """(w, V) = eigh(A) with V constant: backward is V diag(gw) V^H (admm_net.py:303-306)."""

    @staticmethod
    def forward(ctx, A, solver, asm):
        w, V = solver(A.detach())
        ctx.save_for_backward(V)
        ctx.asm = asm
        ctx.mark_non_differentiable(V)
        return w, V

    @staticmethod
    def backward(ctx, gw, _gV):
        (V,) = ctx.saved_tensors
        return ctx.asm.vdvh(V, gw.to(torch.float32)), None, None


class _Rebuild(torch.autograd.Function):
    """G = (G0 + G0^H) / 2 with G0 = V diag(d) V^H, V constant (admm_net.py:336-354); d carries the gradient
    Re(v_c^H S v_c) with S = (g + g^H) / 2, the backward of the symmetrisation followed by that of the two products."""

    @staticmethod
    def forward(ctx, V, d, asm):
        ctx.save_for_backward(V)
        ctx.asm = asm
        return asm.vdvh(V, d)

    @staticmethod
    def backward(ctx, g):
        (V,) = ctx.saved_tensors
        S = 0.5 * (g + g.transpose(1, 2).conj())
        return None, ctx.asm.vhsv(V, S), None


def _block_matrix(phi: torch.Tensor, h: torch.Tensor, corner: float) -> torch.Tensor:
    """[[diag(h), phi], [phi^H, corner]]  (admm_net.py:273-284, 428-439)."""
    B = phi.shape[0]
    top = torch.cat([torch.diag_embed(h).to(phi.dtype), phi.unsqueeze(-1)], dim=2)
    low = torch.cat([phi.conj().unsqueeze(1),
                     torch.full((B, 1, 1), corner, dtype=phi.dtype, device=phi.device)], dim=2)
    return torch.cat([top, low], dim=1)


def _phi_layer(layer, y, b, G, Z):
    """admm_net.py:79-105."""
    rho = F.softplus(layer.rho)
    b_sq = torch.abs(b) ** 2 + EPS
    return b_sq / (1 + rho * b_sq) * (y / (b + EPS) + rho * G[:, :-1, -1] + Z[:, :-1, -1])


def _h_layer(layer, G, Z, sigma):
    """admm_net.py:134-194; returns the diagonal h [B, D]."""
    D = layer.dim
    rho = F.softplus(layer.rho)
    t = (torch.diagonal(G, dim1=1, dim2=2)[:, :D] + torch.diagonal(Z, dim1=1, dim2=2)[:, :D] / (rho + EPS)).real
    A = (2 * torch.sqrt(torch.tensor(float(D))).to(t.device) * sigma + sigma ** 2).reshape(-1, 1)
    tc = t + 0.1 * layer.correction_net(t)
    cval = A * tc.abs().max(dim=1, keepdim=True)[0] + tc.sum(dim=1, keepdim=True)
    scale = torch.clamp(torch.sigmoid(layer.projection_weight) / (cval + EPS), max=1.0)
    return tc * scale


def _g_layer(layer, phi, h, Z, solver, asm):
    """admm_net.py:237-354."""
    corner = (1.0 / (F.softplus(layer.lambda_param) ** 2 + EPS)).item()
    A = _block_matrix(phi, h, corner) - (1.0 / (F.softplus(layer.rho) + EPS)) * Z
    A = 0.5 * (A + A.transpose(1, 2).conj())
    w, V = _EighValuesOnly.apply(A, solver, asm)
    # learned eigenvalue map, every eigenvalue through the same 1 -> 16 -> 1 network (admm_net.py:310-334)
    wp = F.softplus(w - torch.sigmoid(layer.threshold)) * layer.value_net(w.abs().unsqueeze(-1)).squeeze(-1)
    return _Rebuild.apply(V, wp, asm)


def _z_layer(layer, k, phi, h, G, Z):
    """admm_net.py:388-474."""
    corner = (1.0 / (F.softplus(layer.lambda_param) ** 2 + EPS)).item()
    R = G - _block_matrix(phi, h, corner)
    rho = F.softplus(layer.rho)
    rn = torch.linalg.matrix_norm(R)                                   # Frobenius, [B]
    B = rn.shape[0]
    feat = torch.stack([torch.full((B,), k / 10.0, device=rn.device),
                        torch.full((B,), rho.item(), device=rn.device),
                        rn / (rn.mean() + EPS)], dim=1)
    step = rho * (0.5 + 1.5 * layer.residual_scale_net(feat)).squeeze(1)
    return Z + step.reshape(-1, 1, 1) * R


def _peak_head(head, phi):
    """admm_net.py:570-630 (dropout of the attention is live in train mode, as in the reference)."""
    B = phi.shape[0]
    x = head.feature_extractor(torch.cat([phi.real, phi.imag], dim=1))
    pos = head.position_projection(head.position_encoder.unsqueeze(0).expand(B, -1, -1))
    att, _ = head.attention(query=x.unsqueeze(1), key=pos, value=pos)
    xp = head.peak_extractor(x + att.squeeze(1))
    taus, fs, cs = [], [], []
    for t in range(head.L_max):
        z = xp + torch.tensor(t / head.L_max, device=phi.device)
        taus.append(head.tau_regressor[t](z))
        fs.append(head.f_regressor[t](z))
        cs.append(head.confidence_net(z))
    return torch.cat(taus, 1), torch.cat(fs, 1), torch.cat(cs, 1)


def unrolled_forward(model, y: torch.Tensor, b: torch.Tensor, sigma: torch.Tensor,
                     solver: Optional[Callable] = None, assembler=None):
    """Differentiable K-layer forward on the device of ``y`` (admm_net.py:742-764 / 791-816).

    ``solver(A) -> (w, V)`` defaults to the HIP eigensolver and ``assembler`` to the HIP contractions; the CPU unit
    tests pass stand-ins (``torch.linalg.eigh``, ``TorchAssembler``) to check the autograd wiring against the reference's
    gradients without a GPU.
    Returns phi, or (tau, f, confidences, phi) when the model has a PeakSearchLayer.
    """
    solver = ops.eigh if solver is None else solver
    asm = Assembler if assembler is None else assembler
    K, D = model.num_layers, model.M * model.N
    if y.dim() != 2 or y.shape[1] != D or b.shape != y.shape:
        raise ValueError(f"y, b must be [B, {D}] complex; got {tuple(y.shape)}, {tuple(b.shape)}")
    y = y.to(torch.complex64)
    b = b.to(torch.complex64)
    sigma = sigma.to(torch.float32).reshape(-1)
    B, n = y.shape[0], D + 1
    G = torch.zeros(B, n, n, dtype=torch.complex64, device=y.device)
    Z = torch.zeros_like(G)
    phi = None
    for k in range(K):
        phi = _phi_layer(model.phiLayers[k], y, b, G, Z)
        if k == K - 1:
            break
        h = _h_layer(model.hLayers[k], G, Z, sigma)
        G = _g_layer(model.gLayers[k], phi, h, Z, solver, asm)
        Z = _z_layer(model.zLayers[k], k, phi, h, G, Z)
    if getattr(model, "_HAS_HEAD", False):
        tau, f, conf = _peak_head(model.peakSearchLayer, phi)
        return tau, f, conf, phi
    return phi
