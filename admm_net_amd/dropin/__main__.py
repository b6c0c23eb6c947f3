"""``python -m admm_net_amd.dropin script.py [args ...]``: run one of the reference's scripts (main_for_net.py,
test/test_time_net.py, main.py, test/test_time_admm.py, trainPhi.py ...) unchanged on the MI355X path.

The script runs as ``__main__`` with the import order  shims -> repo root -> the script's own directory -> the rest,
exactly what ``python script.py`` gives except that the shims come first.  Nothing is exec'ed: the script runs
inside this interpreter (runpy), so a GPU that is already initialised is not an issue.
"""
import os
import runpy
import sys

from . import activate


def main():
    if len(sys.argv) < 2:
        sys.exit("usage: python -m admm_net_amd.dropin script.py [args ...]")
    script = os.path.abspath(sys.argv[1])
    sys.argv = [script] + sys.argv[2:]
    here = os.path.dirname(script)
    if here in sys.path:
        sys.path.remove(here)
    sys.path.insert(0, here)            # what `python script.py` would have put first ...
    activate()                          # ... and the shims in front of it
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
