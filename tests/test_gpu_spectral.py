"""The G-layer as a matrix function (csrc/spectral.hip, spectral_fused.hip) -- the default route of the forward.

The reference evaluates G = V f(L) V^H through torch.linalg.eigh (/root/reference/admm_net.py:303-354).  The product path
evaluates the same matrix function without an eigendecomposition wherever the spectrum allows it (two outliers + a narrow bulk,
checked per matrix) and sends the remaining matrices through its eigensolver.  These tests pin: the result against the float64
oracle (at least as close as the reference's own float32 arithmetic), that the route is actually taken, that the rejected
matrices are handled, and the run-to-run reproducibility that once failed without the barrier in front of the border rows."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(Nb, Nd, K, B, perturb, seed=5):
    from admm_net_amd import synth
    from oracle import admm_net_ref as R
    torch.set_num_threads(min(8, torch.get_num_threads()))   # (many-core hosts make the small CPU eigh calls crawl)
    sd = R.make_weights(Nb, Nd, K, seed=seed, head=False, perturb=perturb)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=11)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    ref64 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    ref32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
    return sd, (ty, tb, ts), ref64, ref32


@pytest.mark.skipif(os.environ.get("ADMMNET_SPECTRAL") == "0", reason="the route is switched off")
@pytest.mark.parametrize("Nb,Nd,K,B,perturb", [(10, 10, 10, 48, 0.0), (10, 10, 10, 48, 0.3), (8, 16, 8, 32, 0.3),
                                                (16, 16, 8, 12, 0.0), (16, 16, 8, 12, 0.3), (12, 16, 6, 12, 0.3),
                                                (4, 4, 6, 32, 0.3), (5, 7, 6, 16, 0.2), (10, 10, 8, 32, 1.0)])
def test_matrix_function_route_matches_the_f64_oracle(Nb, Nd, K, B, perturb):
    import admm_net_amd as A
    dev = torch.device("cuda:0")
    sd, (ty, tb, ts), ref64, ref32 = _oracle(Nb, Nd, K, B, perturb)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    m.load_state_dict(sd)
    out = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    st = m.last_status
    sc = np.abs(ref64).max()
    err = np.abs(out - ref64).max() / sc
    err32 = np.abs(ref32 - ref64).max() / sc            # what the reference's own float32 forward loses against float64
    assert st[0] == 0
    assert st[1] + st[2] == (K - 2) * B                  # every dense G-layer was either evaluated or handed over (layer 0: arrowhead)
    assert st[2] > 0.5 * (K - 2) * B, st                 # ... and the matrix function is the rule, not the exception
    assert err <= max(3.0 * err32, 2e-5), (err, err32, st)   # float32 tolerance of the parity suite (north_star: 1e-4)
    assert err <= 1e-4


@pytest.mark.skipif(os.environ.get("ADMMNET_SPECTRAL") == "0", reason="the route is switched off")
def test_matrices_with_a_kink_of_the_eigenvalue_map_in_the_bulk_are_handed_to_the_eigensolver():
    """Default-initialised weights at 10 x 10: the ReLU kinks of value_net fall inside the bulk of a good part of the matrices --
    the model check must reject exactly those (status word 3) and the result must not suffer."""
    import admm_net_amd as A
    dev = torch.device("cuda:0")
    sd, (ty, tb, ts), ref64, ref32 = _oracle(10, 10, 10, 64, 0.0)
    m = A.PhiEstADMMNet(M=10, N=10, num_layers=10).eval()
    m.load_state_dict(sd)
    out = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    st = m.last_status
    assert st[3] > 0 and st[1] >= st[3] and st[2] > 0, st
    assert np.abs(out - ref64).max() / np.abs(ref64).max() <= 2e-5


CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import torch
import admm_net_amd as A
from admm_net_amd import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = A.PhiEstADMMNet(M={Nb}, N={Nd}, num_layers={K}).eval()
ty, tb, ts, _ = synth.make_batch_device({B}, {Nb}, {Nd}, seed=20260104, device=dev)
outs = [m(ty, tb, ts).clone() for _ in range(4)]
same = all(torch.equal(outs[0], o) for o in outs[1:])
print("RESULT", int(same), m.last_status)
"""


@pytest.mark.skipif(os.environ.get("ADMMNET_SPECTRAL") == "0", reason="the route is switched off")
@pytest.mark.parametrize("Nb,Nd,K,B", [(16, 16, 6, 8192), (10, 10, 10, 8192), (8, 16, 8, 8192)])
def test_forward_is_bitwise_reproducible_at_scale(Nb, Nd, K, B):
    """Regression: the vector-ALU phase of one wave beside the matrix-core phase of another once made ~1 matrix in 10^4 differ
    from run to run in one border element (csrc/spectral_fused.hip, the barrier in front of the border rows)."""
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, Nb=Nb, Nd=Nd, K=K, B=B)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    assert line.split()[1] == "1", line


@pytest.mark.skipif(os.environ.get("ADMMNET_SPECTRAL") == "0", reason="the route is switched off")
@pytest.mark.parametrize("case", ["y_zero", "y_small", "y_large", "sigma_zero", "sigma_large", "identical", "single", "zeros_in_b"])
@pytest.mark.parametrize("Nb,Nd", [(10, 10), (16, 16)])
def test_degenerate_inputs_are_handed_over_or_evaluated_correctly(Nb, Nd, case):
    """Inputs that break the two-outliers-plus-bulk picture (no signal at all: phi = 0; tiny signals; no noise level; zeros among
    the symbols) must either pass the per-matrix checks honestly or go to the eigensolver -- never a wrong or non-finite result."""
    import admm_net_amd as A
    from admm_net_amd import synth
    from oracle import admm_net_ref as R
    torch.set_num_threads(min(8, torch.get_num_threads()))
    dev = torch.device("cuda:0")
    K, B = 6, 6
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=4)
    if case == "y_zero":
        y = np.zeros_like(y)
    elif case == "y_small":
        y = (y * 1e-3).astype(y.dtype)
    elif case == "y_large":
        y = (y * 1e3).astype(y.dtype)
    elif case == "sigma_zero":
        s = np.zeros_like(s)
    elif case == "sigma_large":
        s = (s * 100).astype(s.dtype)
    elif case == "identical":
        y, b, s = np.repeat(y[:1], B, 0), np.repeat(b[:1], B, 0), np.repeat(s[:1], B, 0)
    elif case == "single":
        y, b, s = y[:1], b[:1], s[:1]
    elif case == "zeros_in_b":
        b = b.copy()
        b[:, ::3] = 0
    sd = R.make_weights(Nb, Nd, K, seed=3, head=False, perturb=0.3)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    ref = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    ref32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    m.load_state_dict(sd)
    out = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    st = m.last_status
    assert np.isfinite(out).all() and st[0] == 0
    assert st[1] + st[2] == (K - 2) * y.shape[0]
    sc = max(np.abs(ref).max(), 1e-30)
    err, e32 = np.abs(out - ref).max() / sc, np.abs(ref32 - ref).max() / sc
    assert err <= max(3.0 * e32, 3e-5), (case, err, e32, st)
    if case == "y_zero":
        assert st[2] == 0      # phi = 0: there is no outlier pair to find -- every matrix must be rejected
