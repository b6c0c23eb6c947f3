"""Developer diagnostic (not collected by pytest): where does the D > 128 pipeline lose accuracy?

    gpurun -- 'python tests/gpu_bisect_cfg3.py > gpurun_out/bisect.log 2>&1'

For every layer of the cfg3-depth case of tests/test_gpu_parity.py::test_cfg3_shape_small_batch the
G-layer is evaluated from IDENTICAL (fp32-rounded) inputs by (a) the HIP path, (b) LAPACK in fp32
(the reference's arithmetic) and (c) float64 (truth for those inputs), and the eigensolver alone is
separated from the rebuild by pushing its (w, V) through a float64 rebuild.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import admm_net_amd as A  # noqa: E402
from admm_net_amd import ops  # noqa: E402
from admm_net_amd.synth import make_batch  # noqa: E402
from oracle import admm_net_ref as R  # noqa: E402

dev = torch.device("cuda:0")
Nb, Nd, K, B = 16, 16, int(os.environ.get("BISECT_K", "16")), int(os.environ.get("BISECT_B", "3"))
sd = R.make_weights(Nb, Nd, K, seed=int(os.environ.get("BISECT_WSEED", "7")), head=False,
                    perturb=float(os.environ.get("BISECT_PERT", "0.3")))
m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
m.load_state_dict(sd)
y, b, s, _ = make_batch(B, Nb, Nd, seed=int(os.environ.get("BISECT_DSEED", "5")))
ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
sd32, sd64 = R.cast_weights(sd, "f32"), R.cast_weights(sd, "f64")


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / np.abs(b).max())


tr = []
o64 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64", trace=tr).numpy()
o32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
phi = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
print(f"end to end: rel(hip, f64) {rel(phi, o64):.3e}   rel(lapack32, f64) {rel(o32, o64):.3e}   "
      f"rel(hip, lapack32) {rel(phi, o32):.3e}", flush=True)

print("layer |  G: hip-vs-f64  lapack32-vs-f64 | eigh only (f64 rebuild): hip   lapack32 | resid hip  lap | orth hip  lap"
      " | spectrum: min max  #in-cluster", flush=True)
for k in range(K - 1):
    t = tr[k]
    phi32 = t["phi"].to(torch.complex64)
    h32 = t["h"].float()
    Z32 = None if k == 0 else tr[k - 1]["Z"].to(torch.complex64)
    Zt = torch.zeros(B, Nb * Nd + 1, Nb * Nd + 1, dtype=torch.complex128) if Z32 is None else Z32.to(torch.complex128)
    Gt, wt, At = R.g_layer(sd64, k, phi32.to(torch.complex128), h32.double(), Zt, return_eig=True)
    Zl = torch.zeros(B, Nb * Nd + 1, Nb * Nd + 1, dtype=torch.complex64) if Z32 is None else Z32
    Gl, wl, Al = R.g_layer(sd32, k, phi32, h32, Zl, return_eig=True)
    Gh, wh, _ = ops.glayer(m, k, phi32.to(dev), h32.to(dev), None if Z32 is None else Z32.to(dev))
    Gh = Gh.cpu()
    # eigensolver alone on the fp32 matrix the reference would hand to LAPACK
    A32 = Al.to(torch.complex64)
    we, Ve = ops.eigh(A32.to(dev))
    we, Ve = we.cpu().double(), Ve.cpu().to(torch.complex128)
    wl2, Vl2 = torch.linalg.eigh(A32)
    wl2, Vl2 = wl2.double(), Vl2.to(torch.complex128)
    A64 = A32.to(torch.complex128)

    def frebuild(w, V):
        f = R.eigenvalue_map(sd64, k, w)
        return V @ torch.diag_embed(f.to(V.dtype)) @ V.transpose(1, 2).conj()
    Gtrue = frebuild(*torch.linalg.eigh(A64))
    nA = float(A64.abs().max())
    res_h = float((A64 @ Ve - Ve * we[:, None, :]).abs().max()) / nA
    res_l = float((A64 @ Vl2 - Vl2 * wl2[:, None, :]).abs().max()) / nA
    eye = torch.eye(A64.shape[1], dtype=torch.complex128)
    or_h = float((Ve.transpose(1, 2).conj() @ Ve - eye).abs().max())
    or_l = float((Vl2.transpose(1, 2).conj() @ Vl2 - eye).abs().max())
    med = wt.median(dim=1).values[:, None]
    ncl = int(((wt - med).abs() < 1e-3 * wt.abs().max()).sum()) // B
    print(f"{k:5d} | {rel(Gh.numpy(), Gt.numpy()):.3e}   {rel(Gl.numpy(), Gt.numpy()):.3e}      | "
          f"{rel(frebuild(we, Ve).numpy(), Gtrue.numpy()):.3e} {rel(frebuild(wl2, Vl2).numpy(), Gtrue.numpy()):.3e} | "
          f"{res_h:.2e} {res_l:.2e} | {or_h:.2e} {or_l:.2e} | {float(wt.min()):.3e} {float(wt.max()):.3e} {ncl}",
          flush=True)

# ---- is the end-to-end difference systematic?  Several weight / data seeds at the same depth: distance to float64 of
#      the HIP path and of the reference arithmetic (LAPACK fp32), per seed.
print("\nseed | rel(hip, f64)  rel(lapack32, f64)  ratio", flush=True)
ratios = []
for seed in range(int(os.environ.get("BISECT_SEEDS", "8"))):
    sds = R.make_weights(Nb, Nd, K, seed=100 + seed, head=False, perturb=0.3 if seed % 2 else 0.0)
    ms = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    ms.load_state_dict(sds)
    ys, bs, ss, _ = make_batch(2, Nb, Nd, seed=200 + seed)
    tys, tbs, tss = torch.from_numpy(ys), torch.from_numpy(bs), torch.from_numpy(ss)
    p_h = ms(tys.to(dev), tbs.to(dev), tss.to(dev)).cpu().numpy()
    p32 = R.forward(sds, tys, tbs, tss, Nb, Nd, K, dtype="f32").numpy()
    p64 = R.forward(sds, tys, tbs, tss, Nb, Nd, K, dtype="f64").numpy()
    eh, el = rel(p_h, p64), rel(p32, p64)
    ratios.append(eh / el)
    print(f"{seed:4d} | {eh:.3e}      {el:.3e}          {eh / el:.2f}", flush=True)
print(f"geometric mean of the ratio: {float(np.exp(np.mean(np.log(ratios)))):.2f}", flush=True)
