"""Loading of the golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).

Two fixture flavours:
  * weights stored (``w:<state_dict key>`` arrays): the reference's own state_dict travels with the case;
  * weights by recipe (``wseed = [seed, perturbed]`` + ``wsum:<key>`` checksums, used for the deep 16x16 cases
    whose state_dict would be 2-4 MB of incompressible floats): the weights are rebuilt by constructing the
    drop-in module under the same ``torch.manual_seed`` (its initial weights are bit-equal to the reference's,
    tests/test_host_logic.py) and applying the same scalar perturbation, then verified against the float64
    per-tensor checksums (sum, sum of squares, first and last element) the generator took from the REFERENCE's tensors.
"""
import numpy as np
import torch


def perturb_(model, seed):
    """The 'perturbed' weight recipe of make_golden.py: scalars ~ N(init, 0.5), attention biases ~ N(0, 0.1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 0:
                p.add_(0.5 * torch.randn((), generator=g))
            elif "in_proj_bias" in name or "out_proj.bias" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def checksum(t):
    a = t.detach().cpu().double().reshape(-1).numpy()
    return np.array([a.sum(), (a * a).sum(), a[0], a[-1], a.size], dtype=np.float64)


def load_fixture(path):
    """-> (npz, state_dict, (Nb, Nd, K, B, L, head, sigma_2d))."""
    z = np.load(path)
    meta = tuple(int(v) for v in z["meta"])
    Nb, Nd, K, B, L, head, s2d = meta
    if "wseed" in z.files:
        import admm_net_amd as A
        seed, perturbed = (int(v) for v in z["wseed"])
        torch.manual_seed(seed)
        m = (A.ADMMNet if head else A.PhiEstADMMNet)(M=Nb, N=Nd, L=L, num_layers=K)
        if perturbed:
            perturb_(m, seed + 1000)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        keys = [k[5:] for k in z.files if k.startswith("wsum:")]
        assert set(keys) == set(sd.keys()), "state_dict key set differs from the reference's"
        for k in keys:
            got, want = checksum(sd[k]), z["wsum:" + k]
            assert np.array_equal(got[2:], want[2:]) and np.allclose(got[:2], want[:2], rtol=1e-12, atol=0), \
                f"rebuilt weight {k} differs from the reference's"
    else:
        sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    return z, sd, meta
