"""Drop-in shims: the reference's module names (``admm_net``, ``admm``, ``utils.peakSearchUtils``, ``utils.mathUtils``)
backed by the MI355X path, plus the mechanism that makes them win over the reference's own files.

The reference scripts are run as ``python main_for_net.py`` from the reference directory, so ``sys.path[0]`` is that
directory and a ``PYTHONPATH`` entry can never shadow its ``admm_net.py``.  Two mechanisms do:

    python -m admm_net_amd.dropin main_for_net.py [args ...]     # launcher: shims first, script directory after
    import admm_net_amd.dropin.activate                           # or: first line of a script / sitecustomize

Both put this directory at the FRONT of ``sys.path``.  ``utils`` here is a package whose ``__path__`` is extended with
the script directory's own ``utils/`` so that un-shimmed submodules (``utils.plotUtils``) still resolve to the
reference's files.
"""
import os
import sys

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(SHIM_DIR))


def activate():
    """Put the shim directory (and the repo root, for ``import admm_net_amd``) at the front of sys.path and drop
    already-imported reference modules of the shimmed names."""
    for p in (REPO_ROOT, SHIM_DIR):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in ("admm_net", "admm", "utils", "utils.peakSearchUtils", "utils.mathUtils"):
        mod = sys.modules.get(name)
        f = getattr(mod, "__file__", None) or ""
        if mod is not None and not os.path.abspath(f).startswith(SHIM_DIR):
            del sys.modules[name]
