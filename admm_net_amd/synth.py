"""Synthetic OFDM-radar scenes shaped like the reference's dataset generator.

Vectorised restatement of the per-sample recipe in
/root/reference/generate_data.py:133-221 (``_generate_single_sample`` and
``_generate_communication_symbols``) built on the signal model helpers of
/root/reference/utils/mathUtils.py:4-21 (vander_vec), :24-50 (kr), :53-68
(pskmod), :71-90 (pskdemod), :93-111 (awgn).  The reference seeds nothing; we
draw from ``numpy.random.default_rng(seed)`` so benchmark and parity inputs
are reproducible.
"""
from __future__ import annotations

import numpy as np

TAU_RANGE = (0.1, 0.9)      # generate_data.py:30
F_RANGE = (-0.4, 0.4)       # generate_data.py:31


def pskmod(data, m, phase_offset=0.0):
    """utils/mathUtils.py:53-68."""
    return np.exp(1j * (2 * np.pi * data / m + phase_offset))


def pskdemod(sig, m, phase_offset=0.0):
    """utils/mathUtils.py:71-90."""
    ang = np.angle(sig) - phase_offset
    ang = np.mod(ang + np.pi / m, 2 * np.pi)
    return np.floor(ang * m / (2 * np.pi)).astype(int) % m


def steering(freq, length):
    """exp(j 2 pi k f), k = 0..length-1  (vander_vec(0, (length-1) f, length))."""
    k = np.arange(length)
    return np.exp(1j * 2 * np.pi * np.multiply.outer(freq, k))


def make_batch(batch, Nb, Nd, L=3, seed=20260104, snr_range=(5.0, 25.0), snr_e=7.0):
    """Return (y [B,D] c64, b [B,D] c64, sigma [B] f32, truth dict).

    D = Nb*Nd, flattening order of ``kr(S, conj(D))``: index = i_b*Nd + i_d.
    """
    rng = np.random.default_rng(seed)
    D = Nb * Nd
    tau = rng.uniform(*TAU_RANGE, size=(batch, L))
    f = rng.uniform(*F_RANGE, size=(batch, L))
    C = rng.normal(0, 0.7, size=(batch, L)) + 1j * rng.normal(0, 0.7, size=(batch, L))
    S = steering(f, Nb)                     # [B, L, Nb]
    Dm = steering(tau, Nd)                  # [B, L, Nd]
    # Psi = kr(S, conj(D)) @ C  -> [B, Nb, Nd] then flattened
    psi = np.einsum("bl,bli,blj->bij", C, S, np.conj(Dm)).reshape(batch, D)
    data = rng.integers(0, 4, size=(batch, D))
    sig = pskmod(data, 4, np.pi / 4)
    p_sig = np.mean(np.abs(sig) ** 2, axis=1, keepdims=True)
    p_noise = p_sig / (10 ** (snr_e / 10))
    sig_n = sig + np.sqrt(p_noise / 2) * (rng.standard_normal((batch, D)) + 1j * rng.standard_normal((batch, D)))
    b = pskmod(pskdemod(sig_n, 4, np.pi / 4), 4, np.pi / 4)
    e = sig - b
    real_y = (b + e) * psi
    snr_w = rng.uniform(*snr_range, size=(batch, 1))
    w = np.sqrt(0.5) * (rng.standard_normal((batch, D)) + 1j * rng.standard_normal((batch, D)))
    w_var = np.sum(np.abs(real_y) ** 2, axis=1, keepdims=True) / (10 ** (snr_w / 10) * D)
    y = real_y + np.sqrt(w_var) * w
    sigma = np.linalg.norm(e / b, axis=1) + 1.0
    truth = dict(tau=tau.astype(np.float32), f=f.astype(np.float32), C=C.astype(np.complex64))
    return y.astype(np.complex64), b.astype(np.complex64), sigma.astype(np.float32), truth


def make_batch_device(batch, Nb, Nd, L=3, seed=20260104, snr_range=(5.0, 25.0), snr_e=7.0, device=None, labels=False,
                      rho=1.0, label_iters=5):
    """The same recipe generated ON the MI355X (csrc/synth.hip, one workgroup per sample): nothing crosses PCIe.

    Returns (y [B,D] c64, b [B,D] c64, sigma [B] f32, truth dict of device tensors); with ``labels=True`` the dict
    carries ``phi`` [B,D] c64 = the label DatasetGeneratorCreatePhi computes with admm_for_us (generate_data.py:410-463).
    The random stream is the kernel's own counter-based generator (the reference seeds nothing), so samples differ from
    ``make_batch`` with the same seed; the distributions are the same.
    """
    import ctypes
    import torch
    from . import _lib
    lib = _lib.load()
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.AdmmNetError("make_batch_device needs the GPU (use make_batch for host data)")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    D = Nb * Nd
    with torch.cuda.device(device):
        y = torch.empty(batch, D, dtype=torch.complex64, device=device)
        b = torch.empty(batch, D, dtype=torch.complex64, device=device)
        sigma = torch.empty(batch, dtype=torch.float32, device=device)
        tau = torch.empty(batch, L, dtype=torch.float32, device=device)
        f = torch.empty(batch, L, dtype=torch.float32, device=device)
        C = torch.empty(batch, L, dtype=torch.complex64, device=device)
        phi = torch.empty(batch, D, dtype=torch.complex64, device=device) if labels else None
        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(lib.admmnet_synth_batch(batch, Nb, Nd, L, int(seed) & (2 ** 64 - 1), float(snr_range[0]),
                                           float(snr_range[1]), float(snr_e), float(rho), int(label_iters), p(y), p(b),
                                           p(sigma), p(tau), p(f), p(C), p(phi), stream), "admmnet_synth_batch")
    truth = dict(tau=tau, f=f, C=C)
    if labels:
        truth["phi"] = phi
    return y, b, sigma, truth
