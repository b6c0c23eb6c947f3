"""Generate golden GRADIENTS by importing the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grads.py
Needs /root/reference (read-only, never copied).  For each case the reference model is put in eval mode
(so the attention dropout is off and the result is deterministic), run with gradients enabled on seeded
inputs, and a fixed linear functional of its outputs is back-propagated.  The fixture holds inputs, the
reference's state_dict, the functional's coefficients, the outputs and d loss / d parameter for every
parameter (``none:<name>`` marks parameters whose ``.grad`` stays None) -- data only, no reference code.
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
from make_golden import perturb_, sd_arrays


def loss_of(out, coef, head):
    """Fixed linear functional: Re <c_phi, phi> (+ <c_tau, tau> + <c_f, f> + <c_conf, conf>)."""
    if head:
        tau, f, conf, phi = out
        return ((coef["c_phi"].conj() * phi).real.sum() + (coef["c_tau"] * tau).sum()
                + (coef["c_f"] * f).sum() + (coef["c_conf"] * conf).sum())
    return (coef["c_phi"].conj() * out).real.sum()


def run_case(name, cls, Nb, Nd, K, B, seed, perturbed):
    torch.manual_seed(seed)
    model = getattr(ref, cls)(M=Nb, N=Nd, L=3, num_layers=K)
    if perturbed:
        perturb_(model, seed + 1000)
    model.eval()
    head = cls == "ADMMNet"
    y, b, sigma, _ = make_batch(B, Nb, Nd, seed=seed + 7)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(sigma)
    g = torch.Generator().manual_seed(seed + 99)
    D = Nb * Nd
    coef = {"c_phi": torch.complex(torch.randn(B, D, generator=g), torch.randn(B, D, generator=g))}
    if head:
        for nm in ("c_tau", "c_f", "c_conf"):
            coef[nm] = torch.randn(B, 3, generator=g)
    out = model(ty, tb, ts)
    loss = loss_of(out, coef, head)
    loss.backward()
    rec = dict(y=y, b=b, sigma=sigma, meta=np.array([Nb, Nd, K, B, 3, int(head), 0]),
               loss=np.float64(loss.item()))
    rec.update({k: v.numpy() for k, v in coef.items()})
    if head:
        rec.update(tau=out[0].detach().numpy(), f=out[1].detach().numpy(), conf=out[2].detach().numpy(),
                   phi=out[3].detach().numpy())
    else:
        rec.update(phi=out.detach().numpy())
    for pname, p in model.named_parameters():
        if p.grad is None:
            rec["none:" + pname] = np.zeros(0)
        else:
            rec["g:" + pname] = p.grad.numpy()
    rec.update(sd_arrays(model))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, loss {loss.item():.6f}")


if __name__ == "__main__":
    run_case("grads_phiest_4x4_K3_perturbed", "PhiEstADMMNet", 4, 4, 3, 5, 21, True)
    run_case("grads_phiest_10x10_K4_default", "PhiEstADMMNet", 10, 10, 4, 3, 22, False)
    run_case("grads_admmnet_3x4_K3_perturbed", "ADMMNet", 3, 4, 3, 4, 23, True)
