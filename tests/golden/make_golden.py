"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (read-only, never copied).  Writes small .npz fixtures
next to this file: inputs, the reference's own state_dict (as arrays) and the
outputs of the reference's forward.  Only data is stored -- no reference code.

Cases:
  * default torch init under torch.manual_seed(s)
  * "perturbed": every scalar parameter ~ N(init, 0.5) so rho / lambda /
    threshold / projection_weight stop being identical across layers
  * tiny sizes (D = 9, 16) carry per-layer phi, h, G, Z; D = 100 / 128 / 256
    carry final outputs only.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import numpy as np
import torch

import admm_net as ref  # the reference module (imported, not copied)
from admm_net_amd.synth import make_batch
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))
from golden_util import checksum, perturb_   # shared with the loader: same perturbation recipe


def sd_arrays(model):
    return {"w:" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


@torch.no_grad()
def run_case(name, cls, Nb, Nd, K, B, seed, perturbed, per_layer, sigma_2d, store_weights=True):
    torch.manual_seed(seed)
    model = getattr(ref, cls)(M=Nb, N=Nd, L=3, num_layers=K)
    if perturbed:
        perturb_(model, seed + 1000)
    model.eval()
    y, b, sigma, _ = make_batch(B, Nb, Nd, seed=seed + 7)
    ty, tb = torch.from_numpy(y), torch.from_numpy(b)
    ts = torch.from_numpy(sigma)
    if sigma_2d:
        ts = ts.reshape(-1, 1)
    out = model(ty, tb, ts)
    rec = dict(y=y, b=b, sigma=sigma, meta=np.array([Nb, Nd, K, B, 3, int(cls == "ADMMNet"), int(sigma_2d)]))
    if cls == "ADMMNet":
        rec.update(tau=out[0].numpy(), f=out[1].numpy(), conf=out[2].numpy(), phi=out[3].numpy())
    else:
        rec.update(phi=out.numpy())
    if per_layer:
        n = Nb * Nd + 1
        G = torch.zeros(B, n, n)
        Z = torch.zeros(B, n, n)
        for k in range(K):
            phi = model.phiLayers[k](ty, tb, G, Z, k)
            H = model.hLayers[k](phi, G, Z, ts, k)
            G = model.gLayers[k](phi, H, Z, k)
            Z = model.zLayers[k](phi, H, G, Z, k)
            rec[f"L{k}:phi"] = phi.numpy()
            rec[f"L{k}:h"] = torch.diagonal(H, dim1=1, dim2=2).numpy()
            rec[f"L{k}:G"] = G.numpy()
            rec[f"L{k}:Z"] = Z.numpy()
        assert torch.equal(phi, out if cls != "ADMMNet" else out[3])
    if store_weights:
        rec.update(sd_arrays(model))
    else:   # deep 16x16 cases: weights by recipe (seed) + checksums of the reference's tensors (golden_util.py)
        rec["wseed"] = np.array([seed, int(perturbed)])
        for k, v in model.state_dict().items():
            rec["wsum:" + k] = checksum(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def split_batch_case():
    """Reference behaviour SURVEY 8(e): the batch mean couples signals."""
    torch.manual_seed(5)
    model = ref.PhiEstADMMNet(M=4, N=4, L=3, num_layers=3).eval()
    y, b, sigma, _ = make_batch(6, 4, 4, seed=55)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(sigma)
    with torch.no_grad():
        full = model(ty, tb, ts).numpy()
        a = model(ty[:3], tb[:3], ts[:3]).numpy()
        c = model(ty[3:], tb[3:], ts[3:]).numpy()
    rec = dict(y=y, b=b, sigma=sigma, phi_full=full, phi_split=np.concatenate([a, c]),
               meta=np.array([4, 4, 3, 6, 3, 0, 0]))
    rec.update(sd_arrays(model))
    np.savez_compressed(os.path.join(HERE, "split_batch_4x4_K3.npz"), **rec)
    print("split_batch: max|full-split| =", np.abs(full - rec["phi_split"]).max())


def geometry_cases():
    """A geometry between the register-resident route (D <= 128) and D = 256: 12 x 16 -> D = 192, the padded route of
    csrc/api.hip (eig_dim).  The reference accepts any M, N (admm_net.py:726-741)."""
    run_case("phiest_12x16_K3_perturbed", "PhiEstADMMNet", 12, 16, 3, 3, 25, True, False, False)


def depth_cases():
    """Fixtures at the depth of the BASELINE configs (cfg2: K = 8 on 8x16; cfg3/4: K = 16 on 16x16; cfg5: K = 32)."""
    run_case("phiest_8x16_K8_perturbed", "PhiEstADMMNet", 8, 16, 8, 3, 21, True, False, False)
    run_case("phiest_16x16_K16_default", "PhiEstADMMNet", 16, 16, 16, 2, 22, False, False, False, store_weights=False)
    run_case("phiest_16x16_K16_perturbed", "PhiEstADMMNet", 16, 16, 16, 2, 23, True, False, True, store_weights=False)
    run_case("phiest_16x16_K32_default", "PhiEstADMMNet", 16, 16, 32, 1, 24, False, False, False, store_weights=False)


if __name__ == "__main__":
    if "--depth-only" in sys.argv:
        depth_cases()
        sys.exit(0)
    if "--geometry-only" in sys.argv:
        geometry_cases()
        sys.exit(0)
    run_case("phiest_3x3_K3_default", "PhiEstADMMNet", 3, 3, 3, 3, 11, False, True, False)
    run_case("phiest_3x3_K3_perturbed", "PhiEstADMMNet", 3, 3, 3, 3, 12, True, True, True)
    run_case("phiest_4x4_K4_perturbed", "PhiEstADMMNet", 4, 4, 4, 3, 13, True, True, False)
    run_case("admmnet_4x4_K3_perturbed", "ADMMNet", 4, 4, 3, 4, 14, True, False, True)
    run_case("phiest_10x10_K5_default", "PhiEstADMMNet", 10, 10, 5, 4, 15, False, False, True)
    run_case("phiest_10x10_K5_perturbed", "PhiEstADMMNet", 10, 10, 5, 4, 16, True, False, False)
    run_case("admmnet_10x10_K3_default", "ADMMNet", 10, 10, 3, 3, 17, False, False, False)
    run_case("phiest_8x16_K3_perturbed", "PhiEstADMMNet", 8, 16, 3, 2, 18, True, False, False)
    run_case("phiest_16x16_K2_default", "PhiEstADMMNet", 16, 16, 2, 2, 19, False, False, False)
    run_case("phiest_16x16_K3_perturbed", "PhiEstADMMNet", 16, 16, 3, 2, 20, True, False, False)
    split_batch_case()
    depth_cases()
    geometry_cases()
