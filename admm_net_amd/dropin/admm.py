"""Shim for ``from admm import *`` in the reference scripts (classical solver, host, no cvxpy needed)."""
import numpy as np  # noqa: F401  (the reference scripts rely on ``from admm import *`` exporting np)
from admm_net_amd.classical import admm_for_us, admm_for_us_G_svd, admm_for_us_H_cvx_0  # noqa: F401
