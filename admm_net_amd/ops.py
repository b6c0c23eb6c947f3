"""Thin torch wrappers over the building-block entry points of include/admmnet.h.

Every function takes CUDA (HIP) tensors, enqueues on the current stream and
returns device tensors.  No CPU path exists.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import Cfg


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.AdmmNetError(f"{name} must be a HIP device tensor (no CPU fallback)")


def eigh(A: torch.Tensor):
    """Batched Hermitian eigendecomposition (torch.linalg.eigh at admm_net.py:303).

    A: [B, n, n] complex64 (lower triangle read).  Returns (w [B, n] unsorted, V [B, n, n]).
    """
    _need_cuda(A, "A")
    lib = _lib.load()
    A = A.to(torch.complex64).contiguous()
    B, n, _ = A.shape
    dev = A.device
    with torch.cuda.device(dev):
        need = lib.admmnet_eigh_workspace_bytes(n, B)
        if need < 0:
            raise _lib.AdmmNetError(f"eigh: unsupported n={n}")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        w = torch.empty(B, n, dtype=torch.float32, device=dev)
        V = torch.empty(B, n, n, dtype=torch.complex64, device=dev)
        status = torch.zeros(4, dtype=torch.int32, device=dev)
        _lib.check(lib.admmnet_eigh_c64(n, B, _ptr(A), _ptr(w), _ptr(V), _ptr(ws), need, _ptr(status),
                                        _stream(dev)), "admmnet_eigh_c64")
        bad = int(status[0].item())
        if bad:
            raise _lib.AdmmNetError(f"eigensolver failed on {bad} matrices")
    return w, V


def glayer(model, k: int, phi: torch.Tensor, h: torch.Tensor, Z=None):
    """GLayer.forward (admm_net.py:237-354) of layer k of ``model`` plus the Z-layer residual norm.

    Returns (G [B,n,n] c64, w [B,n] f32, rn [B] f32).
    """
    _need_cuda(phi, "phi")
    lib = _lib.load()
    dev = phi.device
    B, D = phi.shape
    n = D + 1
    cfg = model.cfg()
    with torch.cuda.device(dev):
        W = model.packed_weights(dev)
        off = lib.admmnet_layer_weight_offset(ctypes.byref(cfg), k)
        lw = W[off:]
        need = lib.admmnet_glayer_workspace_bytes(ctypes.byref(cfg), B)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        G = torch.empty(B, n, n, dtype=torch.complex64, device=dev)
        w = torch.empty(B, n, dtype=torch.float32, device=dev)
        rn = torch.empty(B, dtype=torch.float32, device=dev)
        status = torch.zeros(4, dtype=torch.int32, device=dev)
        phi = phi.to(torch.complex64).contiguous()
        h = h.to(torch.float32).contiguous()
        Zc = None if Z is None else Z.to(torch.complex64).contiguous()
        _lib.check(lib.admmnet_glayer_f32(ctypes.byref(cfg), _ptr(lw), _ptr(phi), _ptr(h), _ptr(Zc), B, _ptr(G),
                                          _ptr(w), _ptr(rn), _ptr(ws), need, _ptr(status), _stream(dev)),
                   "admmnet_glayer_f32")
        bad = int(status[0].item())
        if bad:
            raise _lib.AdmmNetError(f"eigensolver failed on {bad} matrices")
    return G, w, rn


def spectrum(phi: torch.Tensor, xbase: int, ybase: int, taus: torch.Tensor, fs: torch.Tensor):
    """|phi^H kron(s(f), conj d(tau))|^2 on the grid fs x taus (peakSearchUtils.py:9-60), float64.

    phi [B, ybase*xbase] complex64; returns [B, len(fs), len(taus)] float64.
    """
    _need_cuda(phi, "phi")
    lib = _lib.load()
    dev = phi.device
    phi = phi.to(torch.complex64).contiguous()
    B = phi.shape[0]
    taus = taus.to(dev, torch.float64).contiguous()
    fs = fs.to(dev, torch.float64).contiguous()
    nx, ny = taus.numel(), fs.numel()
    with torch.cuda.device(dev):
        need = lib.admmnet_spectrum_workspace_bytes(xbase, ybase, nx, ny)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(B, ny, nx, dtype=torch.float64, device=dev)
        _lib.check(lib.admmnet_spectrum_f64(_ptr(phi), B, xbase, ybase, _ptr(taus), nx, _ptr(fs), ny, _ptr(out),
                                            _ptr(ws), need, _stream(dev)), "admmnet_spectrum_f64")
    return out
