"""Shim for ``from utils.peakSearchUtils import *``: same public names as utils/peakSearchUtils.py -- its own functions
plus everything its ``from utils.mathUtils import *`` / ``import numpy as np`` re-export to star-importers
(test/test_time_net.py:1 gets vander_vec / kr / awgn / pskmod / pskdemod / np this way).  The reference module needs
skimage (absent here); the search itself does not."""
import os
import sys

import numpy as np  # noqa: F401
from .mathUtils import *  # noqa: F401,F403  (the shim next to this file)
from admm_net_amd.peak_search import alt_peak_search, peak_search, peak_search_func, regional_maxima  # noqa: F401


def find_regional_maxima(data, neighborhood_size=3, threshold=None):
    """utils/peakSearchUtils.py:176-194 (defined there, unused by the callers): maximum-filter local maxima."""
    from scipy import ndimage
    data_max = ndimage.maximum_filter(data, size=neighborhood_size)
    maxima = data == data_max
    if threshold is not None:
        maxima = maxima & (data > threshold)
    return maxima


def plot_peaks(func_opts, ground_truth_dict=None, search_opts=None):
    """utils/peakSearchUtils.py:199-324 is a matplotlib figure of the spectrum with the detected / true peaks
    (visualisation: out of scope of the MI355X path).  Kept callable so main.py / main_for_net.py run to their end:
    draws the same quantities in one plain figure."""
    import matplotlib
    if "matplotlib.pyplot" not in sys.modules:
        matplotlib.use(os.environ.get("MPLBACKEND", "Agg"))
    import matplotlib.pyplot as plt
    so = {"xstep": 0.01, "ystep": 0.01, "iter": 1, **(search_opts or {})}
    phi, xb, yb = func_opts["phi"], func_opts["xbase"], func_opts["ybase"]
    ax_x = np.arange(0, 1 - so["xstep"], so["xstep"])
    ax_y = np.arange(-0.5, 0.5 - so["xstep"], so["ystep"])
    X, Y = np.meshgrid(ax_x, ax_y)
    Z = peak_search(phi, X, xb, Y, yb)
    pk = alt_peak_search(func_opts, so)
    fig = plt.figure()
    plt.pcolormesh(X, Y, Z, shading="auto")
    if len(pk):
        plt.plot(pk[:, 0], pk[:, 1], "r+", label="peaks")
    if ground_truth_dict:
        plt.plot(ground_truth_dict["tau"], ground_truth_dict["f"], "wo", mfc="none", label="truth")
    plt.xlabel("tau")
    plt.ylabel("f")
    plt.legend()
    plt.show()
    return fig

