"""Grid peak search on the dual polynomial phi (host driver + optional HIP spectrum).

Mirror of /root/reference/utils/peakSearchUtils.py: ``peak_search_func`` (:9-33),
``peak_search`` (:37-60) and ``alt_peak_search`` (:63-173) with the same
arguments, option keys, defaults, quirks and return layout
(rows [x = tau, y = f, height], float64).

The reference evaluates one kron + dot per grid point in a double Python loop;
here the spectrum uses the separable form  |s^T conj(Phi) conj(d)|^2  with
Phi = phi.reshape(ybase, xbase) (float64, as the reference's complex128), and
``batched_peak_search`` evaluates the coarse grid for a whole batch on the GPU
through ``admmnet_spectrum_f64``.  Regional maxima follow
skimage.morphology.local_maxima(connectivity=2) semantics (8-connected,
plateau aware, borders allowed, a constant image has none); skimage is not
installed in the build image, so this is an own implementation.
"""
from __future__ import annotations

import numpy as np

DEFAULT_OPTS = {"xmin": 0, "xmax": 1, "xstep": 0.01, "ymin": -0.5, "ymax": 0.5, "ystep": 0.01,
                "reducefactor": 0.1, "iter": 1}


def _vander(x, base):
    """utils/mathUtils.py:4-21: exp(j 2 pi linspace(0, (base-1) x, base)) for an array of x."""
    x = np.asarray(x, dtype=np.float64)
    fre = np.linspace(np.zeros_like(x), (base - 1) * x, base, axis=-1)
    return np.exp(1j * 2 * np.pi * fre)


def peak_search_func(phi, x, x_base, y, y_base):
    """peakSearchUtils.py:9-33 for scalar x, y."""
    return float(spectrum_grid(phi, np.array([x]), x_base, np.array([y]), y_base)[0, 0])


def spectrum_grid(phi, xs, x_base, ys, y_base):
    """|phi^H kron(s(y), conj d(x))|^2 for all (y, x) in ys x xs -> [len(ys), len(xs)] float64."""
    phi = np.asarray(phi).reshape(-1).astype(np.complex128)
    Phi = phi.reshape(y_base, x_base)
    S = _vander(ys, y_base)            # [ny, y_base]
    Dm = _vander(xs, x_base)           # [nx, x_base]
    U = np.conj(Phi) @ np.conj(Dm).T   # [y_base, nx]
    return np.abs(S @ U) ** 2


def peak_search(phi, X, x_base, Y, y_base):
    """peakSearchUtils.py:37-60: X, Y are meshgrid arrays of equal shape."""
    xs, ys = X[0, :], Y[:, 0]
    if np.array_equal(X, np.broadcast_to(xs, X.shape)) and np.array_equal(Y, np.broadcast_to(ys[:, None], Y.shape)):
        return spectrum_grid(phi, xs, x_base, ys, y_base)
    out = np.zeros((Y.shape[0], X.shape[1]))
    for i in range(Y.shape[0]):
        for j in range(X.shape[1]):
            out[i, j] = peak_search_func(phi, X[i, j], x_base, Y[i, j], y_base)
    return out


def regional_maxima(img: np.ndarray) -> np.ndarray:
    """8-connected plateau-aware regional maxima (skimage local_maxima(connectivity=2) semantics)."""
    img = np.asarray(img, dtype=np.float64)
    if img.size == 0 or img.ndim != 2:
        return np.zeros(img.shape, dtype=bool)
    if np.all(img == img.flat[0]):
        return np.zeros(img.shape, dtype=bool)   # flat image: the plateau touches the padded border
    H, W = img.shape
    pad = np.full((H + 2, W + 2), -np.inf)
    pad[1:-1, 1:-1] = img
    shifts = [(dy, dx) for dy in (0, 1, 2) for dx in (0, 1, 2) if (dy, dx) != (1, 1)]
    nb = [pad[dy:dy + H, dx:dx + W] for dy, dx in shifts]
    cand = np.ones((H, W), dtype=bool)
    for v in nb:
        cand &= img >= v
    # a plateau is a maximum only if all of its pixels are candidates: propagate rejection
    while True:
        cpad = np.ones((H + 2, W + 2), dtype=bool)
        cpad[1:-1, 1:-1] = cand
        kill = np.zeros((H, W), dtype=bool)
        for (dy, dx), v in zip(shifts, nb):
            kill |= (v == img) & ~cpad[dy:dy + H, dx:dx + W]
        new = cand & ~kill
        if np.array_equal(new, cand):
            return cand
        cand = new


def alt_peak_search(func_opts, opts=None, coarse_Z=None):
    """peakSearchUtils.py:63-173.  ``coarse_Z`` optionally supplies the coarse-grid spectrum
    (e.g. computed on the GPU by ``batched_peak_search``)."""
    so = {**DEFAULT_OPTS, **(opts or {})}
    phi, x_base, y_base = func_opts["phi"], func_opts["xbase"], func_opts["ybase"]
    xmin, xmax, xstep = so["xmin"], so["xmax"], so["xstep"]
    ymin, ymax, ystep = so["ymin"], so["ymax"], so["ystep"]
    reduce_factor, max_iter = so["reducefactor"], so["iter"]
    axis_x = np.arange(xmin, xmax - xstep, xstep)
    axis_y = np.arange(ymin, ymax - xstep, ystep)       # reference quirk: "- xstep" (:106)
    if len(axis_x) == 0 or len(axis_y) == 0:
        return np.zeros((0, 3))
    axis_Z = spectrum_grid(phi, axis_x, x_base, axis_y, y_base) if coarse_Z is None else np.asarray(coarse_Z)
    rows, cols = np.where(regional_maxima(axis_Z))
    num = len(rows)
    res = np.zeros((num, 3))
    res[:, 0] = axis_x[cols]
    res[:, 1] = axis_y[rows]
    lx, ly = xstep, ystep
    for _ in range(max_iter):
        lx = reduce_factor * lx
        ly = reduce_factor * ly
        for k in range(num):
            x0 = max(xmin, res[k, 0] - lx)
            x1 = min(xmax - lx, res[k, 0] + lx)
            y0 = max(ymin, res[k, 1] - ly)
            y1 = min(ymax - ly, res[k, 1] + ly)
            if x0 >= x1 or y0 >= y1:
                continue
            loc_x = np.arange(x0, x1, lx)
            loc_y = np.arange(y0, y1, ly)
            if len(loc_x) == 0 or len(loc_y) == 0:
                continue
            Zl = spectrum_grid(phi, loc_x, x_base, loc_y, y_base)
            zmax = np.max(Zl)
            pr, pc = np.where(Zl == zmax)
            if len(pr) > 0:
                res[k, 0] = loc_x[pc[0]]
                res[k, 1] = loc_y[pr[0]]
                res[k, 2] = zmax
    return res


def coarse_axes(opts=None):
    so = {**DEFAULT_OPTS, **(opts or {})}
    return (np.arange(so["xmin"], so["xmax"] - so["xstep"], so["xstep"]),
            np.arange(so["ymin"], so["ymax"] - so["xstep"], so["ystep"]))


def batched_peak_search(phi, xbase, ybase, opts=None, top=None, max_peaks=256, host_refine=False):
    """Peak search for a batch phi [B, D] (torch tensor on the HIP device).

    Everything runs on the device (``ops.peak_search``: spectrum, regional maxima, refinement rounds, one
    signal per workgroup); only the [B, max_peaks, 3] result comes back.  ``host_refine=True`` keeps the
    earlier split (device spectrum, host maxima + refinement = ``alt_peak_search`` verbatim) for A/B checks.
    Returns a list of [num_peaks, 3] arrays, each sorted by height (descending, stable) and truncated to
    ``top`` rows when given, as the callers do (main_for_net.py:119,126).
    """
    import torch
    from . import ops
    if not host_refine:
        pk, cnt = ops.peak_search(phi, xbase, ybase, opts, max_peaks)
        pk, cnt = pk.cpu().numpy(), cnt.cpu().numpy()
        if int(cnt.max(initial=0)) > max_peaks:
            raise ValueError(f"{int(cnt.max())} regional maxima exceed max_peaks={max_peaks}")
        out = []
        for i in range(pk.shape[0]):
            r = pk[i, :cnt[i]]
            r = r[np.argsort(-r[:, 2], kind="stable")]
            out.append(r[:top] if top else r)
        return out
    ax, ay = coarse_axes(opts)
    Z = ops.spectrum(phi, xbase, ybase, torch.from_numpy(ax), torch.from_numpy(ay)).cpu().numpy()
    phis = phi.detach().cpu().numpy()
    out = []
    for i in range(phis.shape[0]):
        r = alt_peak_search({"phi": phis[i], "xbase": xbase, "ybase": ybase}, opts, coarse_Z=Z[i])
        r = r[np.argsort(-r[:, 2], kind="stable")]
        out.append(r[:top] if top else r)
    return out
