"""Shim for ``from utils.peakSearchUtils import *`` (no skimage / matplotlib needed for the search itself)."""
import numpy as np  # noqa: F401
from admm_net_amd.peak_search import alt_peak_search, peak_search, peak_search_func  # noqa: F401
from admm_net_amd.synth import pskdemod, pskmod  # noqa: F401
