"""Developer aid (GPU box): random geometries / depths / batch sizes / weight perturbations through the default forward against
the float64 oracle.  python tests/gpu_spectral_fuzz.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_net_amd as A
from admm_net_amd import synth
from oracle import admm_net_ref as R

torch.set_num_threads(8)
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
t_start = time.time()
for c in range(cases):
    while True:
        Nb, Nd = int(rng.integers(2, 17)), int(rng.integers(2, 17))
        if Nb * Nd <= 256:
            break
    D = Nb * Nd
    K = int(rng.integers(3, 13 if D <= 128 else 9))
    B = int(rng.integers(1, 33 if D <= 128 else 13))
    pert = float(rng.choice([0.0, 0.2, 0.5, 1.0]))
    seed = int(rng.integers(0, 10 ** 6))
    sd = R.make_weights(Nb, Nd, K, seed=seed, head=False, perturb=pert)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=seed + 1)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    ref = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    ref32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    m.load_state_dict(sd)
    out = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    sc = np.abs(ref).max()
    err, e32 = np.abs(out - ref).max() / sc, np.abs(ref32 - ref).max() / sc
    ok = np.isfinite(out).all() and err <= max(3.0 * e32, 3e-5)
    worst = max(worst, err / max(e32, 1e-7))
    print(f"{'ok  ' if ok else 'FAIL'} {Nb}x{Nd} (D={D}) K={K} B={B} perturb={pert} seed={seed}: err {err:.2e} (f32 oracle {e32:.2e}) "
          f"status={m.last_status}", flush=True)
    if not ok:
        sys.exit(1)
print(f"all {cases} cases ok; worst err / f32-oracle err = {worst:.2f}; {time.time() - t_start:.0f} s")
