"""CPU: the oracle restatement vs the golden vectors generated from the imported reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import admm_net_ref as R

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
CASES = [p for p in GOLD if os.path.basename(p).startswith(("phiest_", "admmnet_"))]
TOL_F32 = 2e-5      # oracle fp32 vs reference fp32: same formulas, different BLAS/LAPACK call order
TOL_F64 = 5e-5      # reference fp32 vs ground truth fp64


from golden_util import load_fixture as load   # weights stored, or rebuilt by seed and checksum-verified


def test_depth_fixtures_present():
    """Reference-made fixtures at the depth of the BASELINE configs (K = 8 on 8x16, K = 16 / 32 on 16x16)."""
    names = {os.path.basename(p)[:-4] for p in CASES}
    assert {"phiest_8x16_K8_perturbed", "phiest_16x16_K16_default", "phiest_16x16_K16_perturbed",
            "phiest_16x16_K32_default"} <= names


def test_fixture_inventory():
    assert len(CASES) >= 10
    assert any("16x16" in p for p in CASES) and any("8x16" in p for p in CASES) and any("10x10" in p for p in CASES)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference(path):
    z, sd, (Nb, Nd, K, B, L, head, s2d) = load(path)
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    for dt, tol in (("f32", TOL_F32), ("f64", TOL_F64)):
        tr = []
        out = R.forward(sd, y, b, s, Nb, Nd, K, L, dtype=dt, head=bool(head), trace=tr)
        phi = (out[3] if head else out).numpy()
        ref = z["phi"]
        assert np.abs(phi - ref).max() <= tol * np.abs(ref).max()
        if head:
            for i, key in enumerate(["tau", "f", "conf"]):
                assert np.abs(out[i].numpy() - z[key]).max() <= 1e-5
        if "L0:phi" in z.files:      # per-layer traces for the tiny cases
            for k in range(K):
                for key in ("phi", "h", "G", "Z"):
                    a, r = tr[k][key].numpy(), z[f"L{k}:{key}"]
                    assert np.abs(a - r).max() <= 5e-5 * max(1.0, np.abs(r).max()), (k, key)


@pytest.mark.parametrize("path", CASES[:4], ids=[os.path.basename(p)[:-4] for p in CASES[:4]])
def test_dead_tail_is_dead(path):
    """admm_net.py:757-764: the last layer's H/G/Z never reach the output."""
    z, sd, (Nb, Nd, K, B, L, head, s2d) = load(path)
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    a = R.forward(sd, y, b, s, Nb, Nd, K, L)
    c = R.forward(sd, y, b, s, Nb, Nd, K, L, skip_dead_tail=True)
    assert torch.equal(a, c)


def test_batch_mean_couples_signals():
    """splitting the batch changes phi (reference measured 1.2e-4 on this fixture)."""
    p = [q for q in GOLD if "split" in q][0]
    z, sd, (Nb, Nd, K, B, L, head, s2d) = load(p)
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    full = R.forward(sd, y, b, s, Nb, Nd, K).numpy()
    split = np.concatenate([R.forward(sd, y[:3], b[:3], s[:3], Nb, Nd, K).numpy(),
                            R.forward(sd, y[3:], b[3:], s[3:], Nb, Nd, K).numpy()])
    assert np.abs(full - z["phi_full"]).max() < 2e-6
    assert np.abs(split - z["phi_split"]).max() < 2e-6
    assert np.abs(full - split).max() > 1e-5


def test_make_weights_keys_match_reference():
    z, sd, (Nb, Nd, K, B, L, head, s2d) = load([p for p in CASES if "admmnet_10x10" in p][0])
    mine = R.make_weights(Nb, Nd, K, L, seed=1, head=True)
    assert set(mine.keys()) == set(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
