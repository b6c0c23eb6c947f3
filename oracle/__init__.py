"""CPU oracle for the ADMM-Net hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker.
The product (``admm_net_amd``) never imports it and fails loudly when its HIP
extension is missing.

Parity status ("Oracle"):
  * ``admm_net_ref``   -- restates /root/reference/admm_net.py; PINNED against
    the imported reference (fixtures under tests/golden/, made by
    tests/golden/make_golden.py in the build container).
  * ``classical_ref``  -- restates admm.py; the reference module cannot be
    imported here (cvxpy absent) and its own tests pin nothing:
    PARITY UNPINNED at the cvxpy/ECOS boundary.
  * ``peak_search_ref``-- restates utils/peakSearchUtils.py; skimage absent and
    no expected outputs ship with the reference: PARITY UNPINNED at the
    skimage.local_maxima boundary (spectrum part is pinned analytically).
"""
