"""Developer diagnostic run on the GPU box (not collected by pytest).

    gpurun -- 'python tests/gpu_probe.py > gpurun_out/probe.log 2>&1'
"""
import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import admm_net_amd as A  # noqa: E402
from admm_net_amd import ops  # noqa: E402
from admm_net_amd.synth import make_batch  # noqa: E402
from oracle import admm_net_ref as R  # noqa: E402

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), flush=True)


def section(s):
    print("\n==== " + s, flush=True)


def eigh_check():
    section("eigh building block vs numpy")
    rng = np.random.default_rng(0)
    for n in [2, 3, 10, 17, 64, 101, 129, 200, 257]:
        Bn = 5
        X = rng.standard_normal((Bn, n, n)) + 1j * rng.standard_normal((Bn, n, n))
        Ah = ((X + X.conj().transpose(0, 2, 1)) / 2).astype(np.complex64)
        t0 = time.time()
        w, V = ops.eigh(torch.from_numpy(Ah).to(dev))
        torch.cuda.synchronize()
        dt = time.time() - t0
        w = w.cpu().numpy().astype(np.float64)
        V = V.cpu().numpy().astype(np.complex128)
        A64 = Ah.astype(np.complex128)
        res = np.abs(A64 @ V - V * w[:, None, :]).max() / np.abs(A64).max()
        orth = np.abs(V.conj().transpose(0, 2, 1) @ V - np.eye(n)).max()
        wr = np.linalg.eigvalsh(A64)
        ee = np.abs(np.sort(w, axis=1) - wr).max() / np.abs(wr).max()
        print(f"n={n:4d} res {res:.2e} orth {orth:.2e} eig {ee:.2e}  ({dt*1e3:.1f} ms)", flush=True)


def load_case(p):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_fixture
    z, sd, (Nb, Nd, K, B, L, head, _) = load_fixture(p)
    cls = A.ADMMNet if head else A.PhiEstADMMNet
    m = cls(M=Nb, N=Nd, L=L, num_layers=K)
    m.load_state_dict(sd)
    m.eval()
    return z, m, sd, (Nb, Nd, K, B, L, head)


def golden_check():
    section("forward vs golden fixtures (reference outputs)")
    for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/*.npz"))):
        name = os.path.basename(p)
        if not name.startswith(("phiest_", "admmnet_")):
            continue
        z, m, sd, (Nb, Nd, K, B, L, head) = load_case(p)
        y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
        try:
            with torch.no_grad():
                out = m(y, b, s)
        except Exception as e:  # noqa: BLE001
            print(f"{name:36s} FAILED: {e}", flush=True)
            continue
        phi = out[3] if head else out
        ref = z["phi"]
        f64 = R.forward(sd, y, b, s, Nb, Nd, K, L, dtype="f64", head=bool(head))
        p64 = (f64[3] if head else f64).numpy()
        e_ref = np.abs(phi.numpy() - ref).max() / np.abs(ref).max()
        e_64 = np.abs(phi.numpy() - p64).max() / np.abs(p64).max()
        r_64 = np.abs(ref - p64).max() / np.abs(p64).max()
        msg = f"{name:36s} |hip-ref| {e_ref:.2e}  |hip-f64| {e_64:.2e}  |ref-f64| {r_64:.2e}"
        if head:
            msg += "  head " + " ".join(f"{np.abs(out[i].numpy() - z[k]).max():.1e}" for i, k in enumerate(["tau", "f", "conf"]))
        print(msg, flush=True)


def glayer_check():
    section("glayer building block vs oracle (layer traces)")
    p = os.path.join(ROOT, "tests/golden/phiest_8x16_K3_perturbed.npz")
    z, m, sd, (Nb, Nd, K, B, L, head) = load_case(p)
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    tr = []
    R.forward(sd, y, b, s, Nb, Nd, K, L, dtype="f64", trace=tr)
    Zprev = None
    for k, t in enumerate(tr):
        phi = t["phi"].to(torch.complex64).to(dev)
        h = t["h"].to(torch.float32).to(dev)
        Zin = None if k == 0 else tr[k - 1]["Z"].to(torch.complex64).to(dev)
        G, w, rn = ops.glayer(m, k, phi, h, Zin)
        Gr = t["G"].numpy()
        eG = np.abs(G.cpu().numpy() - Gr).max() / np.abs(Gr).max()
        ew = np.abs(np.sort(w.cpu().numpy(), 1) - t["w"].numpy()).max() / np.abs(t["w"].numpy()).max()
        ern = np.abs(rn.cpu().numpy() - t["rn"].numpy()).max() / np.abs(t["rn"].numpy()).max()
        print(f"layer {k}: G rel {eG:.2e}  eig rel {ew:.2e}  rn rel {ern:.2e}", flush=True)


def timing():
    section("timing")
    for (Nb, Nd, K, B) in [(10, 10, 10, 1024), (8, 16, 8, 4096), (16, 16, 4, 256)]:
        torch.manual_seed(0)
        m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
        y, b, s, _ = make_batch(B, Nb, Nd, seed=3)
        y, b, s = torch.from_numpy(y).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)
        with torch.no_grad():
            m(y, b, s)
            torch.cuda.synchronize()
            t0 = time.time()
            m(y, b, s)
            torch.cuda.synchronize()
            dt = time.time() - t0
        n = Nb * Nd + 1
        F = R.flops_per_signal(K, n, Nb * Nd)
        print(f"{Nb}x{Nd} K={K} B={B}: {dt*1e3:.1f} ms  {B/dt:.0f} signals/s  {F*B/dt/1e12:.2f} TF/s ({F*B/dt/157.3e12*100:.2f}% of fp32 MFMA peak)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["eigh", "glayer", "golden", "timing"]
    for w in which:
        try:
            {"eigh": eigh_check, "glayer": glayer_check, "golden": golden_check, "timing": timing}[w]()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print("SECTION FAILED:", w, e, flush=True)
