"""Training path: gradients of the differentiable forward against the
gradients of the imported reference (tests/golden/grads_*.npz, written by make_golden_grads.py).

CPU part (not gpu): the autograd wiring -- layer formulas, detached corners / rho feature, eigenvalue-only
gradient, dead last layer -- with ``torch.linalg.eigh`` handed in as a stand-in eigensolver.
GPU part: the same comparison through the product path, i.e. with the HIP eigensolver under the
custom autograd function, via ``model.train(); model(y, b, sigma)`` exactly as trainPhi.py / train.py call it.

Tolerance: gradients are fp32 sums over the whole unrolled graph; the reference and a float64 evaluation
of the same graph differ by ~1e-5 relative, so 5e-4 * max|grad| per parameter (+ 1e-6 absolute) is used.
"""
import glob
import os

import numpy as np
import pytest
import torch

import admm_net_amd as A
from admm_net_amd import _lib, training

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "grads_*.npz")))
RTOL = 5e-4


def load(path, device="cpu"):
    z = np.load(path)
    Nb, Nd, K, B, L, head, _ = [int(v) for v in z["meta"]]
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = (A.ADMMNet if head else A.PhiEstADMMNet)(M=Nb, N=Nd, L=L, num_layers=K)
    m.load_state_dict(sd)
    m = m.to(device)
    t = lambda k: torch.from_numpy(z[k]).to(device)
    return z, m, head, t


def loss_of(out, t, head):
    if head:
        tau, f, conf, phi = out
        return ((t("c_phi").conj() * phi).real.sum() + (t("c_tau") * tau).sum() + (t("c_f") * f).sum()
                + (t("c_conf") * conf).sum())
    return (t("c_phi").conj() * out).real.sum()


def check_grads(z, m, loss):
    assert abs(loss.item() - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    worst = 0.0
    for name, p in m.named_parameters():
        if "none:" + name in z.files:
            assert p.grad is None, f"{name}: the reference leaves this gradient None"
            continue
        want = z["g:" + name]
        assert p.grad is not None, name
        got = p.grad.detach().cpu().numpy()
        err = np.abs(got - want).max()
        tol = RTOL * np.abs(want).max() + 1e-6
        worst = max(worst, err / tol)
        assert err <= tol, f"{name}: |dgrad| {err:.3e} > {tol:.3e}"
    return worst


def cpu_eigh(Amat):
    return torch.linalg.eigh(Amat)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_autograd_wiring_matches_reference_gradients(path):
    z, m, head, t = load(path)
    m.eval()   # fixture convention: dropout off
    out = training.unrolled_forward(m, t("y"), t("b"), t("sigma"), solver=cpu_eigh, assembler=training.TorchAssembler)
    phi = out[3] if head else out
    assert np.abs(phi.detach().numpy() - z["phi"]).max() <= 2e-5 * np.abs(z["phi"]).max()
    loss = loss_of(out, t, head)
    loss.backward()
    check_grads(z, m, loss)


def test_train_mode_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = A.PhiEstADMMNet(M=3, N=3, num_layers=2)      # .train() by default
    with pytest.raises(_lib.AdmmNetError):
        m(torch.zeros(1, 9, dtype=torch.complex64), torch.ones(1, 9, dtype=torch.complex64), torch.ones(1))


def test_rebuild_function_has_the_gradient_of_the_two_products():
    """``_Rebuild`` (V diag(d) V^H + symmetrisation as ONE function with a hand-written backward) against autograd
    through the reference's own formulation -- two matmuls and the symmetrisation (admm_net.py:336-354)."""
    torch.manual_seed(0)
    B, n = 3, 7
    X = torch.randn(B, n, n, dtype=torch.complex128)
    V = torch.linalg.eigh(X + X.transpose(1, 2).conj())[1]
    d1 = torch.randn(B, n, dtype=torch.float64, requires_grad=True)
    d2 = d1.detach().clone().requires_grad_(True)
    c = torch.randn(B, n, n, dtype=torch.complex128)
    G1 = training._Rebuild.apply(V, d1, training.TorchAssembler)
    G0 = torch.matmul(torch.matmul(V, torch.diag_embed(d2).to(V.dtype)), V.transpose(1, 2).conj())
    G2 = 0.5 * (G0 + G0.transpose(1, 2).conj())
    assert torch.allclose(G1, G2, atol=1e-12)
    (c.conj() * G1).real.sum().backward()
    (c.conj() * G2).real.sum().backward()
    assert torch.allclose(d1.grad, d2.grad, atol=1e-11), (d1.grad - d2.grad).abs().max()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_assembly_kernels_match_their_definitions():
    """admmnet_vdvh_c64 / admmnet_vhsv_f32 (csrc/vdvh.hip) against float64 evaluations of V diag(d) V^H and
    Re(v_c^H S v_c), at the reference's n = 101, at a ragged small size and at n = 257; the assembled matrix is exactly
    Hermitian."""
    from admm_net_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    for n, B in ((101, 5), (7, 3), (257, 2), (33, 4)):
        X = torch.randn(B, n, n, dtype=torch.complex128, generator=g)
        V = torch.linalg.eigh(X + X.transpose(1, 2).conj())[1]
        d = torch.randn(B, n, dtype=torch.float64, generator=g)
        Y = torch.randn(B, n, n, dtype=torch.complex128, generator=g)
        S = Y + Y.transpose(1, 2).conj()
        want = torch.matmul(V * d.unsqueeze(1), V.transpose(1, 2).conj())
        got = ops.vdvh(V.to(dev, torch.complex64), d.to(dev, torch.float32)).cpu().to(torch.complex128)
        assert (got - want).abs().max() <= 2e-6 * want.abs().max() * (n ** 0.5)
        assert torch.equal(got, got.transpose(1, 2).conj())
        wq = (V.conj() * torch.matmul(S, V)).sum(dim=1).real
        gq = ops.vhsv(V.to(dev, torch.complex64), S.to(dev, torch.complex64)).cpu().to(torch.float64)
        assert (gq - wq).abs().max() <= 2e-6 * wq.abs().max() * (n ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_hip_training_gradients_match_reference(path):
    assert torch.cuda.is_available(), "needs the MI355X"
    z, m, head, t = load(path, "cuda:0")
    if head:
        m.eval()                                          # attention dropout off, as in the fixture
        out = m.forward_autograd(t("y"), t("b"), t("sigma"))
    else:
        m.train()                                         # the call trainPhi.py makes
        out = m(t("y"), t("b"), t("sigma"))
    phi = out[3] if head else out
    assert phi.requires_grad and phi.is_cuda
    assert np.abs(phi.detach().cpu().numpy() - z["phi"]).max() <= 1e-4 * np.abs(z["phi"]).max()
    loss = loss_of(out, t, head)
    loss.backward()
    check_grads(z, m, loss)


@pytest.mark.gpu
def test_hip_training_step_reduces_loss_and_matches_inference():
    """A few AdamW steps (train.py:13-443 uses AdamW + grad clipping) lower a phi-regression loss, and the
    train-mode forward agrees with the fused inference forward on the same weights."""
    assert torch.cuda.is_available()
    from admm_net_amd import synth
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = A.PhiEstADMMNet(M=4, N=4, num_layers=3).to(dev)
    y, b, sigma, _ = synth.make_batch(16, 4, 4, seed=5)
    ty, tb, ts = (torch.from_numpy(v).to(dev) for v in (y, b, sigma))
    target = ty / tb                                     # any fixed regression target will do
    m.train()
    phi_train = m(ty, tb, ts)
    m.eval()
    with torch.no_grad():
        phi_eval = m(ty, tb, ts)
    assert (phi_train - phi_eval).abs().max().item() <= 1e-4 * phi_eval.abs().max().item()
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = (m(ty, tb, ts) - target).abs().pow(2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
