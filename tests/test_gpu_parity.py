"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the oracle, the golden
fixtures generated from the reference, and size-independent properties at the BASELINE sizes.

Tolerances (fp32 path, stated per north_star):
  * phi: max-abs error <= 1e-4 * max|phi| against the reference fixtures.  The reference itself sits
    6e-7 .. 7e-6 from the float64 evaluation of its own formulas (tests/test_oracle_golden.py); the HIP
    path is additionally required to be no further than 3x the reference's own distance + 2e-6.
    At the cfg3 depth (K = 16, n = 257) fp32 round-off of ANY implementation reaches 1e-4: tolerance 5e-4.
  * G (one layer): 2e-5 relative; eigenvalues 1e-5 relative to the spectral radius.
  * spectrum (float64): 1e-10 relative; peak indices bit-exact.
"""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

import admm_net_amd as A
from admm_net_amd import _lib, ops, peak_search, sharded, synth
from oracle import admm_net_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = sorted(p for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")) if "split" not in p)
TOL_PHI = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib.load()
    return torch.device("cuda:0")


def load_case(p):
    z = np.load(p)
    Nb, Nd, K, B, L, head, s2d = [int(v) for v in z["meta"]]
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = (A.ADMMNet if head else A.PhiEstADMMNet)(M=Nb, N=Nd, L=L, num_layers=K)
    m.load_state_dict(sd)
    return z, m.eval(), sd, (Nb, Nd, K, B, L, head, s2d)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


# ------------------------------------------------------------------ building blocks
@pytest.mark.parametrize("n", [2, 3, 10, 17, 65, 101, 129, 130, 169, 200, 229, 256, 257])
def test_eigh_block(dev, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((6, n, n)) + 1j * rng.standard_normal((6, n, n))
    Ah = ((X + X.conj().transpose(0, 2, 1)) / 2).astype(np.complex64)
    Ah[5] = np.diag(rng.standard_normal(n)).astype(np.complex64)          # already diagonal: tau = 0 path
    w, V = ops.eigh(torch.from_numpy(Ah).to(dev))
    w, V = w.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.complex128)
    A64 = Ah.astype(np.complex128)
    assert np.abs(A64 @ V - V * w[:, None, :]).max() < 3e-5 * np.abs(A64).max()
    assert np.abs(V.conj().transpose(0, 2, 1) @ V - np.eye(n)).max() < 3e-5
    assert np.abs(np.sort(w, 1) - np.linalg.eigvalsh(A64)).max() < 1e-5 * np.abs(A64).max() * max(1, n / 32)


def test_eigh_graded_and_clustered(dev):
    """(scalar I + tiny diagonal + large low rank): the shape of the real layer matrices, where a QL
    sweep in the wrong direction stalls -- exercises the direction choice of the QL kernel."""
    n, rng = 129, np.random.default_rng(5)
    mats = []
    for scale in (1.0, 50.0, 1000.0):
        U = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
        M = 0.003 * np.eye(n) + np.diag(1e-4 * rng.standard_normal(n)) + scale * U @ np.diag([5.0, -7.0, 2.0]) @ U.conj().T
        mats.append((M + M.conj().T) / 2)
        mats.append(mats[-1][::-1, ::-1].copy())                             # graded the other way
    Ah = np.stack(mats).astype(np.complex64)
    w, V = ops.eigh(torch.from_numpy(Ah).to(dev))
    w, V = w.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.complex128)
    A64 = Ah.astype(np.complex128)
    rec = (V * w[:, None, :]) @ V.conj().transpose(0, 2, 1)
    for i in range(len(mats)):
        assert np.abs(rec[i] - A64[i]).max() < 3e-5 * np.abs(A64[i]).max()
        assert np.abs(V[i].conj().T @ V[i] - np.eye(n)).max() < 3e-5


def test_glayer_block_vs_oracle(dev):
    z, m, sd, (Nb, Nd, K, B, L, head, _) = load_case(os.path.join(ROOT, "tests/golden/phiest_8x16_K3_perturbed.npz"))
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    tr = []
    R.forward(sd, y, b, s, Nb, Nd, K, L, dtype="f64", trace=tr)
    for k, t in enumerate(tr):
        Zin = None if k == 0 else tr[k - 1]["Z"].to(torch.complex64).to(dev)
        G, w, rn = ops.glayer(m, k, t["phi"].to(torch.complex64).to(dev), t["h"].float().to(dev), Zin)
        G = G.cpu().numpy()
        assert rel(G, t["G"].numpy()) < 2e-5
        assert np.array_equal(G, G.conj().transpose(0, 2, 1))               # exactly Hermitian, as admm_net.py:352
        assert np.all(G[:, np.arange(G.shape[1]), np.arange(G.shape[1])].imag == 0)
        assert rel(np.sort(w.cpu().numpy(), 1), t["w"].numpy()) < 1e-5
        assert rel(rn.cpu().numpy(), t["rn"].numpy()) < 1e-5


@pytest.mark.parametrize("fixture", ["phiest_8x16_K3_perturbed", "phiest_16x16_K3_perturbed"],
                         ids=["fused_D128", "global_image_D256"])
def test_glayer_first_layer_arrowhead_edge_cases(dev, fixture):
    """Layer 0 (Z = 0) runs the direct arrowhead eigensolver (arrow.hip; D <= 128 fused with the rebuild, larger D
    through the global eigenvector image): repeated h (rotation deflation), zero phi entries (trivial deflation),
    all-equal h, strong coupling -- against the oracle's G-layer in f64."""
    z, m, sd, (Nb, Nd, K, B, L, head, _) = load_case(os.path.join(ROOT, f"tests/golden/{fixture}.npz"))
    D = Nb * Nd
    rng = np.random.default_rng(17)
    phis, hs = [], []
    for case in range(6):
        h = rng.uniform(0.05, 1.0, D)
        p = (rng.standard_normal(D) + 1j * rng.standard_normal(D)) * 0.1
        if case == 1: h = np.round(h, 2)
        if case == 2: p[::3] = 0
        if case == 3: h[:] = 0.37
        if case == 4: p *= 40
        if case == 5: h = np.sort(h); h[10:30] = h[10]; p[50:70] *= 1e-7
        phis.append(p); hs.append(h)
    phi = torch.from_numpy(np.stack(phis)).to(torch.complex128)
    h = torch.from_numpy(np.stack(hs)).to(torch.float64)
    sd64 = {k_: (v.double() if v.is_floating_point() else v) for k_, v in sd.items()}
    Zero = torch.zeros(phi.shape[0], D + 1, D + 1, dtype=torch.complex128)
    Gref, wref, _ = R.g_layer(sd64, 0, phi, h, Zero, return_eig=True)
    G, w, rn = ops.glayer(m, 0, phi.to(torch.complex64).to(dev), h.float().to(dev), None)
    G = G.cpu().numpy()
    assert rel(G, Gref.numpy()) < 2e-5
    assert np.array_equal(G, G.conj().transpose(0, 2, 1))
    assert rel(np.sort(w.cpu().numpy(), 1), wref.numpy()) < 1e-5
    _, rn_ref, _ = R.z_layer(sd64, 0, phi, h, Gref, Zero, return_aux=True)   # residual norm of the Z-layer (its own corner)
    assert rel(rn.cpu().numpy(), rn_ref.numpy()) < 2e-5


# ------------------------------------------------------------------ whole forward vs the reference
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_forward_matches_reference_fixture(dev, path):
    z, m, sd, (Nb, Nd, K, B, L, head, s2d) = load_case(path)
    y, b = torch.from_numpy(z["y"]), torch.from_numpy(z["b"])
    s = torch.from_numpy(z["sigma"])
    if s2d:
        s = s.reshape(-1, 1)                                                  # both shapes occur
    out = m(y, b, s)                                                          # CPU in, CPU out; no no_grad: as main_for_net.py
    phi = out[3] if head else out
    assert phi.device.type == "cpu" and phi.dtype == torch.complex64 and not phi.requires_grad
    ref = z["phi"]
    e_ref = rel(phi.numpy(), ref)
    assert e_ref < TOL_PHI
    p64 = R.forward(sd, y, b, s, Nb, Nd, K, L, dtype="f64", head=bool(head))
    p64 = (p64[3] if head else p64).numpy()
    assert rel(phi.numpy(), p64) <= 3 * rel(ref, p64) + 2e-6
    if head:
        for i, key in enumerate(["tau", "f", "conf"]):
            assert out[i].shape == (B, L) and out[i].dtype == torch.float32
            assert np.abs(out[i].numpy() - z[key]).max() < 2e-5


@pytest.mark.parametrize("shape", [(10, 10, 10, 12, 0.0), (10, 10, 5, 9, 0.5), (8, 16, 8, 10, 0.5), (16, 16, 4, 4, 0.5),
                                   (1, 1, 3, 5, 0.5), (2, 1, 2, 1, 0.0), (5, 7, 1, 9, 0.5)])
def test_forward_vs_oracle_seeded(dev, shape):
    Nb, Nd, K, B, pert = shape
    sd = R.make_weights(Nb, Nd, K, seed=Nb * 100 + K, head=True, perturb=pert)
    m = A.ADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    missing = m.load_state_dict(sd, strict=True)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=B)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    tau, f, conf, phi = m(ty.to(dev), tb.to(dev), ts.to(dev))
    assert phi.is_cuda
    o32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32", head=True)
    o64 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64", head=True)
    assert rel(phi.cpu().numpy(), o32[3].numpy()) < TOL_PHI
    assert rel(phi.cpu().numpy(), o64[3].numpy()) <= 3 * rel(o32[3].numpy(), o64[3].numpy()) + 2e-6
    for a, r in zip((tau, f, conf), o32[:3]):
        assert np.abs(a.cpu().numpy() - r.numpy()).max() < 5e-5


def test_chunking_is_invisible(dev):
    """Eigensolver work chunks (workspace reuse) must not change anything."""
    Nb, Nd, K, B = 4, 5, 3, 37
    torch.manual_seed(3)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=1)
    args = [torch.from_numpy(v).to(dev) for v in (y, b, s)]
    m.chunk = 0
    ref = m(*args).cpu()
    m.chunk = 8
    m._ws = None
    assert torch.equal(m(*args).cpu(), ref)


def test_layer_api_and_sharded_single_rank(dev):
    Nb, Nd, K, B = 6, 6, 4, 20
    torch.manual_seed(4)
    m = A.ADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=2)
    args = [torch.from_numpy(v).to(dev) for v in (y, b, s)]
    tau, f, conf, phi = m(*args)
    sphi, shead = sharded.ShardedForward(m, scope="global")(*args)
    assert torch.equal(sphi, phi) and torch.equal(shead[0], tau) and torch.equal(shead[2], conf)
    # 'shard' scope on two halves == two independent forwards (reference semantics on sub-batches)
    h = B // 2
    a = m(*[t[:h] for t in args])[3]
    c = sharded.ShardedForward(m, scope="shard")(*[t[:h] for t in args])[0]
    assert torch.equal(a, c)


@pytest.mark.parametrize("geom", [(8, 16, 4), (16, 16, 3)], ids=["8x16", "16x16"])
def test_degenerate_inputs_match_oracle(dev, geom):
    """Edge inputs of the data domain: an all-zero observation (phi = 0: the arrowhead layer deflates everything,
    later layers see diagonal + corner matrices), a single non-zero sample, a batch of one, identical signals
    (equal residual norms in the batch mean), a huge and a tiny overall scale."""
    Nb, Nd, K = geom
    torch.manual_seed(6)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    y, b, s, _ = synth.make_batch(6, Nb, Nd, seed=9)
    y[0] = 0
    y[1] = 0
    y[1, 5] = 1.0 + 0.5j
    y[2] = y[3]
    b[2] = b[3]
    s[2] = s[3]
    y[4] *= 1e3
    y[5] *= 1e-4
    for sl in (slice(0, 6), slice(0, 1)):
        ty, tb, ts = (torch.from_numpy(v[sl]) for v in (y, b, s))
        ref = R.forward(sd, ty, tb, ts, Nb, Nd, K, 3, dtype="f64").numpy()
        got = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
        assert np.isfinite(got).all()
        # per signal: the scales differ by 1e7 across the batch
        for i in range(got.shape[0]):
            den = max(np.abs(ref[i]).max(), 1e-30)
            assert np.abs(got[i] - ref[i]).max() <= 2e-4 * den + 1e-12, (i, np.abs(got[i] - ref[i]).max(), den)


def test_fails_loudly_on_nonfinite_input(dev):
    m = A.PhiEstADMMNet(M=3, N=3, num_layers=3).eval()
    y, b, s, _ = synth.make_batch(2, 3, 3, seed=1)
    y[0, 0] = np.nan
    with pytest.raises(_lib.AdmmNetError):
        m(torch.from_numpy(y).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev))


def test_bad_shapes_raise(dev):
    m = A.PhiEstADMMNet(M=3, N=3, num_layers=2).eval()
    with pytest.raises(ValueError):
        m(torch.zeros(2, 8, dtype=torch.complex64), torch.zeros(2, 8, dtype=torch.complex64), torch.ones(2))
    with pytest.raises(_lib.AdmmNetError):
        ops.eigh(torch.zeros(1, 3, 3, dtype=torch.complex64))               # CPU tensor: no fallback


# ------------------------------------------------------------------ properties at the BASELINE sizes
def test_cfg2_full_batch_properties(dev):
    """BASELINE cfg 2 (K=8, D=128, B=4096): too big for the oracle in seconds -> properties.
    (i) a sub-batch evaluated by the oracle with the full-batch means injected agrees;
    (ii) permuting the batch permutes the output (the only coupling is the batch mean)."""
    Nb, Nd, K, B = 8, 16, 8, 4096
    torch.manual_seed(0)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=20260104)
    ty, tb, ts = (torch.from_numpy(v).to(dev) for v in (y, b, s))
    phi = m(ty, tb, ts)
    assert torch.isfinite(torch.view_as_real(phi)).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    phi_p = m(ty[perm], tb[perm], ts[perm])
    assert rel(phi_p.cpu().numpy(), phi[perm].cpu().numpy()) < 2e-5
    # oracle on 6 signals, batch means taken from a layer-wise HIP run of the full batch
    eng = sharded.HipLayerEngine(m, ty, tb, ts)
    eng.begin()
    means = []
    for k in range(K):
        sc = eng.front(k)
        if k == K - 1:
            break
        means.append(float(sc[0].item()) / B)
        eng.back(k, sc[0] / sc[1])
    phi2, _ = eng.finish()
    assert torch.equal(phi2, phi)
    sd = {k_: v.detach() for k_, v in m.state_dict().items()}
    idx = torch.tensor([0, 1, 777, 2048, 4000, 4095])
    o = R.forward(sd, torch.from_numpy(y)[idx], torch.from_numpy(b)[idx], torch.from_numpy(s)[idx], Nb, Nd, K,
                  dtype="f64", skip_dead_tail=True,
                  mean_norm_fn=lambda k, rn: torch.tensor(means[k], dtype=rn.dtype))
    assert rel(phi[idx.to(dev)].cpu().numpy(), o.numpy()) < TOL_PHI


def test_cfg3_shape_small_batch(dev):
    """BASELINE cfg 3/4/5 geometry (D=256, n=257, K=16) on a batch the oracle finishes in seconds."""
    Nb, Nd, K, B = 16, 16, 16, 3
    sd = R.make_weights(Nb, Nd, K, seed=7, head=False, perturb=0.3)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    m.load_state_dict(sd)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=5)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    phi = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    o32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
    o64 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    # 15 dense 257x257 eigen-functions deep, fp32 arithmetic itself is the limit: the reference's own
    # fp32 evaluation sits 4e-5 .. 1e-4 (seed dependent) from the float64 value of the same formulas, and
    # two correct fp32 implementations land at different points of that cloud.  Stated tolerance for
    # this depth: 5e-4 relative, against both the fp32 restatement and the float64 ground truth.
    TOL_DEEP = 5e-4
    ref_err = rel(o32, o64)
    assert ref_err < TOL_DEEP                      # sanity of the yardstick itself
    assert rel(phi, o64) <= TOL_DEEP
    assert rel(phi, o32) <= TOL_DEEP


# ------------------------------------------------------------------ spectrum / peak search
def test_spectrum_and_peak_indices(dev):
    Nb = Nd = 10
    y, b, s, truth = synth.make_batch(4, Nb, Nd, seed=11, snr_range=(20.0, 20.0))
    torch.manual_seed(1)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=4).eval()
    phi = m(torch.from_numpy(y).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev))
    opts = {"xstep": 1 / (10 * Nd), "ystep": 1 / (10 * Nb), "iter": 3}        # main_for_net.py:112-116
    ax, ay = peak_search.coarse_axes(opts)
    Zg = ops.spectrum(phi, Nb, Nd, torch.from_numpy(ax), torch.from_numpy(ay)).cpu().numpy()
    ph = phi.cpu().numpy()
    for i in range(4):
        Zh = peak_search.spectrum_grid(ph[i], ax, Nb, ay, Nd)
        assert np.abs(Zg[i] - Zh).max() < 1e-10 * Zh.max()
        assert np.array_equal(peak_search.regional_maxima(Zg[i]), peak_search.regional_maxima(Zh))   # indices bit-exact
    got = peak_search.batched_peak_search(phi, Nb, Nd, opts, top=3)
    for i in range(4):
        host = peak_search.alt_peak_search({"phi": ph[i], "xbase": Nb, "ybase": Nd}, opts)
        host = host[np.argsort(-host[:, 2], kind="stable")][:3]
        assert np.array_equal(got[i][:, :2], host[:, :2])
        assert np.abs(got[i][:, 2] - host[:, 2]).max() < 1e-9 * host[:, 2].max()


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(10, 10, 1), (8, 16, 2), (16, 16, 1)], ids=["10x10", "8x16", "16x16"])
def test_device_peak_search_matches_host(dev, geom):
    """ops.peak_search (spectrum + regional maxima + refinement on the device, peaks.hip) against the host
    mirror of alt_peak_search on the same phi: same maxima in the same (np.where) order, coordinates bit-exact,
    heights to float64 rounding."""
    Nb, Nd, iters = geom
    y, b, s, _ = synth.make_batch(6, Nb, Nd, seed=23, snr_range=(10.0, 20.0))
    torch.manual_seed(2)
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=3).eval()
    phi = m(torch.from_numpy(y).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev))
    opts = {"xstep": 1 / (4 * Nd), "ystep": 1 / (4 * Nb), "iter": iters}
    pk, cnt = ops.peak_search(phi, Nb, Nd, opts, max_peaks=512)
    pk, cnt = pk.cpu().numpy(), cnt.cpu().numpy()
    ph = phi.cpu().numpy()
    for i in range(ph.shape[0]):
        host = peak_search.alt_peak_search({"phi": ph[i], "xbase": Nb, "ybase": Nd}, opts)
        assert cnt[i] == host.shape[0] and cnt[i] > 0
        got = pk[i, :cnt[i]]
        assert np.array_equal(got[:, :2], host[:, :2])
        assert np.abs(got[:, 2] - host[:, 2]).max() <= 1e-9 * host[:, 2].max()
        assert not pk[i, cnt[i]:].any()
