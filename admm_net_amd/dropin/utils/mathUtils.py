"""Shim for ``from utils.mathUtils import *``: signal-model helpers the demo scripts call (same five names as
utils/mathUtils.py plus ``np``, which star-importers of the reference module also receive)."""
import numpy as np


def vander_vec(x, y, length):
    """utils/mathUtils.py:4-21."""
    return np.exp(1j * 2 * np.pi * np.linspace(x, y, length)).reshape(-1, 1)


def kr(A, B):
    """utils/mathUtils.py:24-50 (column-wise Kronecker product)."""
    if A.shape[1] != B.shape[1]:
        raise ValueError("column counts differ")
    return (A[:, None, :] * B[None, :, :]).reshape(A.shape[0] * B.shape[0], A.shape[1]).astype(complex)


def awgn(sig, snr):
    """utils/mathUtils.py:93-111."""
    p = np.mean(np.abs(sig) ** 2) / (10 ** (snr / 10))
    return sig + np.sqrt(p / 2) * (np.random.randn(len(sig)) + 1j * np.random.randn(len(sig)))


from admm_net_amd.synth import pskdemod, pskmod  # noqa: E402,F401
