"""``utils`` package of the drop-in: peakSearchUtils / mathUtils come from the shims in this directory, every other
submodule (plotUtils ...) from the ``utils/`` directory of the script that is being run (the reference's own)."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in list(sys.path):
    _cand = os.path.join(_p or os.getcwd(), "utils")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != _here and _cand not in __path__:
        __path__.append(_cand)
