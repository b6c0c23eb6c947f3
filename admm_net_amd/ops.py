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


def vdvh(V: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """out = V diag(d) V^H, exactly Hermitian (admm_net.py:336-354; backward of the eigenvalue-only eigh, :303-306).

    V: [B, n, n] complex64 (columns = eigenvectors), d: [B, n] float32.  Returns [B, n, n] complex64."""
    _need_cuda(V, "V")
    lib = _lib.load()
    V = V.to(torch.complex64).contiguous()
    d = d.to(device=V.device, dtype=torch.float32).contiguous()
    B, n, _ = V.shape
    if d.shape != (B, n):
        raise ValueError(f"d must be [{B}, {n}], got {tuple(d.shape)}")
    with torch.cuda.device(V.device):
        out = torch.empty_like(V)
        _lib.check(lib.admmnet_vdvh_c64(n, B, _ptr(V), _ptr(d), _ptr(out), _stream(V.device)), "admmnet_vdvh_c64")
    return out


def vhsv(V: torch.Tensor, S: torch.Tensor) -> torch.Tensor:
    """q[b, c] = Re(v_c^H S v_c) for Hermitian S (lower triangle read): the adjoint of ``vdvh`` with respect to d.

    V, S: [B, n, n] complex64.  Returns [B, n] float32."""
    _need_cuda(V, "V")
    lib = _lib.load()
    V = V.to(torch.complex64).contiguous()
    S = S.to(device=V.device, dtype=torch.complex64).contiguous()
    B, n, _ = V.shape
    if S.shape != V.shape:
        raise ValueError("S must have the shape of V")
    with torch.cuda.device(V.device):
        q = torch.empty(B, n, dtype=torch.float32, device=V.device)
        _lib.check(lib.admmnet_vhsv_f32(n, B, _ptr(V), _ptr(S), _ptr(q), _stream(V.device)), "admmnet_vhsv_f32")
    return q


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


def peak_search(phi: torch.Tensor, xbase: int, ybase: int, opts=None, max_peaks: int = 256):
    """alt_peak_search (utils/peakSearchUtils.py:63-173) for a whole batch on the device: coarse spectrum,
    regional maxima and the refinement rounds, one signal per workgroup.

    Returns (peaks [B, max_peaks, 3] float64 = (tau, f, height) in np.where order of the coarse maxima,
    counts [B] int32 = number of regional maxima; rows >= counts[b] are zero, counts above max_peaks mean
    the list was truncated).
    """
    from . import peak_search as ps
    import ctypes as _ct
    _need_cuda(phi, "phi")
    lib = _lib.load()
    dev = phi.device
    B, D = phi.shape
    if D != xbase * ybase:
        raise ValueError(f"phi has {D} entries, expected xbase*ybase = {xbase * ybase}")
    so = {**ps.DEFAULT_OPTS, **(opts or {})}
    ax, ay = ps.coarse_axes(opts)
    nx, ny = len(ax), len(ay)
    with torch.cuda.device(dev):
        peaks = torch.zeros(B, max_peaks, 3, dtype=torch.float64, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        if nx == 0 or ny == 0:
            return peaks, counts
        tx = torch.from_numpy(ax).to(dev)
        ty = torch.from_numpy(ay).to(dev)
        need = lib.admmnet_peak_search_workspace_bytes(xbase, ybase, nx, ny, B)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        o7 = (_ct.c_double * 7)(so["xmin"], so["xmax"], so["xstep"], so["ymin"], so["ymax"], so["ystep"],
                               so["reducefactor"])
        phi = phi.to(torch.complex64).contiguous()
        _lib.check(lib.admmnet_peak_search_f64(_ptr(phi), B, xbase, ybase, _ptr(tx), nx, _ptr(ty), ny, o7,
                                               int(so["iter"]), max_peaks, _ptr(peaks), _ptr(counts), _ptr(ws), need,
                                               _stream(dev)), "admmnet_peak_search_f64")
    return peaks, counts


def regional_maxima(Z: torch.Tensor):
    """skimage.morphology.local_maxima(connectivity=2) (utils/peakSearchUtils.py:118) of a batch of images on the
    device (the maxima stage of the peak-search kernel).  Z [B, ny, nx] float64 -> bool mask [B, ny, nx]."""
    _need_cuda(Z, "Z")
    lib = _lib.load()
    dev = Z.device
    Z = Z.to(torch.float64).contiguous()
    B, ny, nx = Z.shape
    cap = nx * ny
    with torch.cuda.device(dev):
        peaks = torch.zeros(B, cap, 3, dtype=torch.float64, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        _lib.check(lib.admmnet_regional_maxima_f64(_ptr(Z), B, nx, ny, cap, _ptr(peaks), _ptr(counts), _stream(dev)),
                   "admmnet_regional_maxima_f64")
    mask = torch.zeros(B, ny, nx, dtype=torch.bool, device=dev)
    cnt = counts.cpu()
    for i in range(B):
        k = int(cnt[i])
        if k:
            mask[i, peaks[i, :k, 1].long(), peaks[i, :k, 0].long()] = True
    return mask
