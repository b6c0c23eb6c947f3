"""Property tests (test strategy (v); hypothesis), CPU only: host logic and the oracle's own
invariants -- the statements the GPU parity tests rely on without spelling them out.

* shard bounds tile the batch; the sharded protocol with scope="global" is the reference on the whole batch whatever
  the split (that one is in test_sharded_gloo.py with real ranks);
* regional maxima: the host mirror equals the oracle's flood fill on images FULL of plateaus (few grey levels), is
  invariant to a strictly increasing remap of the grey levels and to mirroring the image, and a constant image has none
  (the semantics of skimage.morphology.local_maxima(connectivity=2) as used at utils/peakSearchUtils.py:118);
* the G-layer consumes only V f(L) V^H (admm_net.py:258): that product is invariant to the phase of every eigenvector and
  to any unitary mixing inside a degenerate eigenspace -- which is why the HIP path is free to return different
  eigenvectors than LAPACK and parity is asserted on G / phi, never on V;
* the oracle's forward is equivariant under permutations of the batch (the only coupling is the batch MEAN of
  admm_net.py:459, which a permutation leaves unchanged up to the order of a float32 sum).
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from admm_net_amd import peak_search, sharded, synth
from oracle import admm_net_ref as R
from oracle import peak_search_ref

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow], derandomize=True)


@settings(max_examples=200, **COMMON)
@given(total=st.integers(0, 5000), world=st.integers(1, 16))
def test_shard_bounds_tile_the_batch(total, world):
    b = [sharded.shard_bounds(total, world, r) for r in range(world)]
    assert b[0][0] == 0 and b[-1][1] == total
    assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes   # as even as possible, big shards first


@settings(max_examples=150, **COMMON)
@given(h=st.integers(1, 9), w=st.integers(1, 9), levels=st.integers(1, 4), seed=st.integers(0, 10 ** 6))
def test_regional_maxima_equal_the_flood_fill_on_plateau_images(h, w, levels, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, levels, size=(h, w)).astype(np.float64)
    got = peak_search.regional_maxima(img)
    assert np.array_equal(got, peak_search_ref.regional_maxima_floodfill(img))
    if levels == 1:
        assert not got.any()                                       # a constant image has no regional maximum
    remap = np.cumsum(rng.random(levels) + 0.1)                     # strictly increasing grey-level map
    assert np.array_equal(peak_search.regional_maxima(remap[img.astype(int)]), got)
    assert np.array_equal(peak_search.regional_maxima(img[::-1, ::-1])[::-1, ::-1], got)
    assert np.array_equal(peak_search.regional_maxima(img.T).T, got)


@settings(max_examples=40, **COMMON)
@given(n=st.integers(2, 12), seed=st.integers(0, 10 ** 6), mult=st.integers(1, 3))
def test_eigen_function_is_invariant_to_the_choice_of_eigenvectors(n, seed, mult):
    mult = min(mult, n)
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(X)
    lam = np.sort(rng.standard_normal(n))
    lam[:mult] = lam[0]                                            # an eigenvalue of multiplicity `mult`
    f = np.log1p(np.exp(lam - 0.3)) * (1.0 / (1.0 + np.exp(-np.abs(lam))))   # any function of the eigenvalue
    G = (Q * f) @ Q.conj().T
    phases = np.exp(2j * np.pi * rng.random(n))
    U, _ = np.linalg.qr(rng.standard_normal((mult, mult)) + 1j * rng.standard_normal((mult, mult)))
    Q2 = Q * phases
    Q2[:, :mult] = Q2[:, :mult] @ U                                 # unitary mixing inside the degenerate eigenspace
    G2 = (Q2 * f) @ Q2.conj().T
    assert np.abs(G - G2).max() <= 1e-12 * max(1.0, np.abs(G).max())
    assert np.abs(G - G.conj().T).max() <= 1e-13 * max(1.0, np.abs(G).max())


@settings(max_examples=6, **COMMON)
@given(seed=st.integers(0, 1000))
def test_oracle_forward_is_equivariant_under_batch_permutations(seed):
    Nb, Nd, K, B = 3, 4, 3, 5
    sd = R.make_weights(Nb, Nd, K, seed=seed, head=False, perturb=0.3)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=seed + 1)
    perm = np.random.default_rng(seed).permutation(B)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    out = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    outp = R.forward(sd, ty[perm], tb[perm], ts[perm], Nb, Nd, K, dtype="f64").numpy()
    assert np.abs(outp - out[perm]).max() <= 1e-10 * np.abs(out).max()
