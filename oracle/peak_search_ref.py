"""Literal CPU restatement of the grid peak search (TEST INFRASTRUCTURE ONLY).

/root/reference/utils/peakSearchUtils.py:9-33 (one kron + dot per grid point),
:37-60 (double loop) and :63-173 (alt_peak_search).  The regional maxima call
into skimage (:118, not installed, version unpinned) is replaced by a
definition-level flood fill: PARITY UNPINNED at that boundary; the reference's
plateau example (:427-432) has no recorded expected mask.
"""
import numpy as np


def vander_vec(x, y, length):
    """utils/mathUtils.py:4-21."""
    return np.exp(1j * 2 * np.pi * np.linspace(x, y, length)).reshape(-1, 1)


def peak_search_func(phi, x, x_base, y, y_base):
    """peakSearchUtils.py:9-33."""
    s = vander_vec(0, (y_base - 1) * y, y_base)
    d = vander_vec(0, (x_base - 1) * x, x_base)
    a = np.kron(s, np.conj(d))
    return float(np.abs(np.dot(np.asarray(phi).conj().T, a).item()) ** 2)


def peak_search(phi, X, x_base, Y, y_base):
    """peakSearchUtils.py:37-60."""
    out = np.zeros((Y.shape[0], X.shape[1]))
    for i in range(Y.shape[0]):
        for j in range(X.shape[1]):
            out[i, j] = peak_search_func(phi, X[i, j], x_base, Y[i, j], y_base)
    return out


def regional_maxima_floodfill(img):
    """Connected sets of equal pixels (8-neighbourhood) all of whose outside neighbours are strictly
    lower; image borders allowed; a constant image has no maximum (skimage pads with the minimum and
    rejects plateaus touching the pad)."""
    img = np.asarray(img, dtype=float)
    H, W = img.shape
    out = np.zeros((H, W), dtype=bool)
    seen = np.zeros((H, W), dtype=bool)
    if np.all(img == img[0, 0]):
        return out
    for i in range(H):
        for j in range(W):
            if seen[i, j]:
                continue
            v = img[i, j]
            stack, comp, is_max = [(i, j)], [], True
            seen[i, j] = True
            while stack:
                a, c = stack.pop()
                comp.append((a, c))
                for da in (-1, 0, 1):
                    for dc in (-1, 0, 1):
                        if da == 0 and dc == 0:
                            continue
                        p, q = a + da, c + dc
                        if p < 0 or q < 0 or p >= H or q >= W:
                            continue
                        if img[p, q] > v:
                            is_max = False
                        elif img[p, q] == v and not seen[p, q]:
                            seen[p, q] = True
                            stack.append((p, q))
            if is_max:
                for a, c in comp:
                    out[a, c] = True
    return out


def alt_peak_search_literal(func_opts, opts=None):
    """peakSearchUtils.py:63-173 as written."""
    d = {'xmin': 0, 'xmax': 1, 'xstep': 0.01, 'ymin': -0.5, 'ymax': 0.5, 'ystep': 0.01, 'reducefactor': 0.1, 'iter': 1}
    so = {**d, **(opts or {})}
    phi, x_base, y_base = func_opts['phi'], func_opts['xbase'], func_opts['ybase']
    xmin, xmax, xstep = so['xmin'], so['xmax'], so['xstep']
    ymin, ymax, ystep = so['ymin'], so['ymax'], so['ystep']
    rf, iters = so['reducefactor'], so['iter']
    ax = np.arange(xmin, xmax - xstep, xstep)
    ay = np.arange(ymin, ymax - xstep, ystep)
    if len(ax) == 0 or len(ay) == 0:
        return np.zeros((0, 3))
    X, Y = np.meshgrid(ax, ay)
    Z = peak_search(phi, X, x_base, Y, y_base)
    xp, yp = np.where(regional_maxima_floodfill(Z))
    num = len(xp)
    res = np.zeros((num, 3))
    for i in range(num):
        res[i, 0] = X[xp[i], yp[i]]
        res[i, 1] = Y[xp[i], yp[i]]
    lx, ly = xstep, ystep
    for _ in range(iters):
        lx, ly = rf * lx, rf * ly
        for k in range(num):
            x0 = max(xmin, res[k, 0] - lx); x1 = min(xmax - lx, res[k, 0] + lx)
            y0 = max(ymin, res[k, 1] - ly); y1 = min(ymax - ly, res[k, 1] + ly)
            if x0 >= x1 or y0 >= y1:
                continue
            lxs = np.arange(x0, x1, lx); lys = np.arange(y0, y1, ly)
            if len(lxs) == 0 or len(lys) == 0:
                continue
            LX, LY = np.meshgrid(lxs, lys)
            LZ = peak_search(phi, LX, x_base, LY, y_base)
            m = np.max(LZ)
            pos = np.where(LZ == m)
            if len(pos[0]) > 0:
                res[k, 0] = LX[pos[0][0], pos[1][0]]
                res[k, 1] = LY[pos[0][0], pos[1][0]]
                res[k, 2] = m
    return res
